// single_scale.hip -- the single-scale trained model as ONE call: what depth_estimation_opticalflow.lua:66-116 runs per frame pair for a
// model that is not multiscale --
//   filter:forward(frame) of both frames                      getFilter, opticalflow_model.lua:45-79 (conv [tanh] conv ...)
//   prepareInput(geometry, last_im, im)                       :131-151: patch 1 narrowed by the search window (rows ceil(maxh/2).., H - maxh + 1 of them)
//   model:forward(input), getModel(geometry, true, true)      :81-129: SpatialMatching(maxh, maxw) -> Minus -> SoftMax over the window
//   processOutput(geometry, moutput, true [, threshold])      :201-252: arg-max with the centre tie-break (or extractOutput(p, 0.11) and
//                                                             scores > threshold), x2yx minus centered2onebased(0, 0), centre paste into
//                                                             full [2][hImg][wImg] and full_confidences [hImg][wImg]
// Module by module (network.py getModel + opticalflow_model.py processOutput) the matcher writes the H1 x W1 x maxh x maxw volume (291 MB
// for a VGA pair at 16 x 16), Minus and the soft-max re-stream it and processOutput streams it once more.  Here the window never leaves
// the CU: the flat-tile matcher's soft-max epilogue (feat_matching_flat.hip, FF_SOFT) does the per-pixel tail on the costs it has just
// summed.  Shapes that kernel does not take go through the stand-alone device ops in the same order (the same arithmetic: every soft-max
// on the device is softmin_body's), so the results are the module path's bit for bit either way.
#include "dfe_internal.h"

namespace {

// out [C][Hc][Wc] = in [C][H][W] rows y0.., columns x0..
__global__ __launch_bounds__(256) void ss_crop_kernel(const float *__restrict__ in, long long plane, int W, int y0, int x0, int Hc, int Wc, long long total,
                                                      float *__restrict__ out) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long row = e / Wc;
        const int x = (int)(e - row * Wc);
        const long long c = row / Hc;
        const int y = (int)(row - c * Hc);
        out[e] = in[c * plane + (long long)(y0 + y) * W + x0 + x];
    }
}

// the index's decode and the centre paste of processOutput (opticalflow_model.lua:207-250) for the path that went through the volume
__global__ __launch_bounds__(256) void ss_paste_kernel(const long long *__restrict__ idx, const float *__restrict__ scores, int H1, int W1, int maxh, int maxw,
                                                       int use_thr, float thr, int ho, int wo, int wFull, long long fullplane, float *__restrict__ full,
                                                       float *__restrict__ full_conf) {
    const long long P = (long long)H1 * W1;
    const int yoff = (maxh + 1) / 2, xoff = (maxw + 1) / 2;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(p / W1), x = (int)(p - (long long)y * W1);
        const int i0 = (int)idx[p] - 1, ty = i0 / maxw;
        const long long fo = (long long)(ho + y) * wFull + wo + x;
        if (full) {
            full[fo] = (float)(ty + 1 - yoff);
            full[fullplane + fo] = (float)(i0 - ty * maxw + 1 - xoff);
        }
        if (full_conf) full_conf[fo] = use_thr ? (scores[p] > thr ? 1.f : 0.f) : 1.f;
    }
}

// torch.Tensor(2, hImg, wImg):zero() / full_confidences:zero() (opticalflow_model.lua:236,243) where the paste does not write: the frame
// around the pasted region (two hipMemsetAsync of the whole planes cost 10 us of a 200-us step)
__global__ __launch_bounds__(256) void ss_border_kernel(float *__restrict__ full, float *__restrict__ full_conf, int hImg, int wImg, int ho, int wo, int H1, int W1) {
    const long long P = (long long)hImg * wImg;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(p / wImg), x = (int)(p - (long long)y * wImg);
        if (y >= ho && y < ho + H1 && x >= wo && x < wo + W1) continue;
        if (full) { full[p] = 0.f; full[P + p] = 0.f; }
        if (full_conf) full_conf[p] = 0.f;
    }
}

// imaxs = middle, scores = 0 where extractOutput will write nothing (the reference passes uninitialised tensors: SURVEY appendix A)
__global__ __launch_bounds__(256) void ss_fill_kernel(long long *__restrict__ idx, float *__restrict__ scores, long long P, long long middle) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        idx[p] = middle;
        scores[p] = 0.f;
    }
}

int ss_grid(long long n) {
    const long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : b > 16384 ? 16384 : b);
}

}  // namespace

extern "C" int dfe_flow_pair_filtered_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, const dfe_filter_layer *layers, int nlayers,
                                          int maxh, int maxw, int use_threshold, double threshold, int hImg, int wImg, float *full, float *full_conf,
                                          int64_t *index, float *scores) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, I0 && I1 && (nlayers == 0 || layers), DFE_E_ARG, "dfe_flow_pair_filtered_f32: NULL argument");
    DFE_REQUIRE(ctx, full || full_conf || index || scores, DFE_E_ARG, "dfe_flow_pair_filtered_f32: no output requested");
    DFE_REQUIRE(ctx, C > 0 && H > 0 && W > 0 && nlayers >= 0 && nlayers <= 8 && maxh > 0 && maxw > 0, DFE_E_ARG,
                "dfe_flow_pair_filtered_f32: C=%d %dx%d, %d layers, window %dx%d", C, H, W, nlayers, maxh, maxw);
    int hk = 1, wk = 1, K = C, maxplanes = C;
    for (int i = 0; i < nlayers; ++i) {
        DFE_REQUIRE(ctx, layers[i].weight && layers[i].kH > 0 && layers[i].kW > 0 && layers[i].nIn > 0 && layers[i].nOut > 0, DFE_E_ARG,
                    "dfe_flow_pair_filtered_f32: layer %d is incomplete", i);
        DFE_REQUIRE(ctx, (i == 0 ? layers[0].nIn == C : (layers[i].conn || layers[i].nIn == layers[i - 1].nOut)), DFE_E_SHAPE,
                    "dfe_flow_pair_filtered_f32: layer %d reads %d planes, its input has %d", i, layers[i].nIn, i == 0 ? C : layers[i - 1].nOut);
        hk += layers[i].kH - 1;
        wk += layers[i].kW - 1;
        K = layers[i].nOut;
        maxplanes = maxplanes > K ? maxplanes : K;
    }
    const int Hf = H - hk + 1, Wf = W - wk + 1;                     // the feature maps (in2 of the matcher)
    const int H1 = Hf - maxh + 1, W1 = Wf - maxw + 1;               // the model's output region = the narrowed in1
    DFE_REQUIRE(ctx, H1 > 0 && W1 > 0, DFE_E_SHAPE, "dfe_flow_pair_filtered_f32: frame %dx%d too small for window %dx%d behind a %dx%d filter", H, W, maxh, maxw, hk, wk);
    DFE_REQUIRE(ctx, !(full || full_conf) || (hImg >= H1 && wImg >= W1), DFE_E_SHAPE, "dfe_flow_pair_filtered_f32: full frame %dx%d smaller than the output %dx%d", hImg, wImg,
                H1, W1);
    const int ny = (maxh + 1) / 2 - 1, nx = (maxw + 1) / 2 - 1;     // prepareInput: narrow(2, ceil(maxh/2), ..) 1-based -> first row / column, 0-based
    const int N = maxh * maxw;
    const long long P1 = (long long)H1 * W1;
    const bool lean = dfe_feat_matching_flat_argmin_takes(ctx, K, H1, W1, maxh, maxw);
    // arena: cropped frame 0 | two ping-pong feature buffers per branch | (fallback) contiguous in1 | volume | probabilities | index | scores
    const int Hc = H1 + hk - 1, Wc = W1 + wk - 1;                   // the part of frame 0 the narrowed features come from
    auto al = [](size_t f) { return (f + 63) / 64 * 64; };
    const size_t f_c0 = nlayers ? (size_t)C * Hc * Wc : 0;
    const size_t f_fa = nlayers ? (size_t)maxplanes * Hc * Wc : 0, f_fb = nlayers ? (size_t)maxplanes * H * W : 0;
    const size_t f_in1 = (!lean && !nlayers) ? (size_t)K * P1 : 0;
    const size_t f_vol = lean ? 0 : (size_t)P1 * N;
    const size_t f_idx = (lean || index) ? 0 : (size_t)P1 * 2, f_sc = (lean || scores || !use_threshold) ? 0 : (size_t)P1;
    void *scr = nullptr;
    int rc = dfe_scratch(ctx, (al(f_c0) + 2 * al(f_fa) + 2 * al(f_fb) + al(f_in1) + 2 * al(f_vol) + al(f_idx) + al(f_sc)) * sizeof(float), &scr, nlayers > 0);
    if (rc) return rc;
    float *c0 = (float *)scr;
    float *fa[2] = {c0 + al(f_c0), c0 + al(f_c0) + al(f_fa)};
    float *fb[2] = {fa[1] + al(f_fa), fa[1] + al(f_fa) + al(f_fb)};
    float *in1c = fb[1] + al(f_fb), *vol = in1c + al(f_in1), *prob = vol + al(f_vol);
    long long *idx_s = (long long *)(prob + al(f_vol));
    float *sc_s = (float *)idx_s + al(f_idx);
    const float *in1 = nullptr, *in2 = nullptr;
    int pitch1 = W1;
    long long plane1 = P1;
    if (nlayers) {
        DfeStageScope st(ctx, DFE_STAGE_FILTER);
        // the first branch filters only what the narrow keeps: a convolution is 'valid' and shift-invariant, so the features of the cropped
        // frame ARE the narrowed features of the whole frame, bit for bit
        // -- and the first layer reads that part of frame 0 in place (rows W apart, planes H * W apart) where the batched kernel takes the
        // layer; otherwise it is copied out first
        const float *ia = I0 + (long long)ny * W + nx, *ib = I1;
        int ha = Hc, wa = Wc, hb = H, wb = W;
        for (int i = 0; i < nlayers; ++i) {
            const float *in_[2] = {ia, ib};
            const dfe_filter_layer *L2[2] = {&layers[i], &layers[i]};
            const int H2[2] = {ha, hb}, W2[2] = {wa, wb};
            float *o2[2] = {fa[i & 1], fb[i & 1]};
            bool viewed = false;
            if (i == 0) {
                const int pit[2] = {W, W};
                const long long pla[2] = {(long long)H * W, (long long)H * W};
                rc = dfe_filter_layer_forward_batch_view(ctx, 2, in_, L2, H2, W2, pit, pla, o2, &viewed);
                if (rc) return rc;
                if (!viewed) {
                    hipLaunchKernelGGL(ss_crop_kernel, dim3(ss_grid((long long)C * Hc * Wc)), dim3(256), 0, ctx->stream, I0, (long long)H * W, W, ny, nx, Hc, Wc,
                                       (long long)C * Hc * Wc, c0);
                    DFE_LAUNCH_CHECK(ctx);
                    in_[0] = c0;
                }
            }
            if (!viewed) rc = dfe_filter_layer_forward_batch(ctx, 2, in_, L2, H2, W2, o2);
            if (rc) return rc;
            ia = o2[0]; ib = o2[1];
            ha -= layers[i].kH - 1; wa -= layers[i].kW - 1; hb -= layers[i].kH - 1; wb -= layers[i].kW - 1;
        }
        DFE_REQUIRE(ctx, ha == H1 && wa == W1 && hb == Hf && wb == Wf, DFE_E_SHAPE, "dfe_flow_pair_filtered_f32: internal shape mismatch");
        in1 = ia; in2 = ib;
    } else {
        // geometry.prefilter: the caller ran the filter (depth_estimation_opticalflow.lua:66-75) -- patch 1's narrow is a view of its map
        in1 = I0 + (long long)ny * W + nx; pitch1 = W; plane1 = (long long)H * W;
        in2 = I1;
    }
    const int ho = (hImg - H1) / 2, wo = (wImg - W1) / 2;
    if ((full || full_conf) && (hImg > H1 || wImg > W1)) {
        hipLaunchKernelGGL(ss_border_kernel, dim3(ss_grid((long long)hImg * wImg)), dim3(256), 0, ctx->stream, full, full_conf, hImg, wImg, ho, wo, H1, W1);
        DFE_LAUNCH_CHECK(ctx);
    }
    if (lean) {
        DfeStageScope st(ctx, DFE_STAGE_MATCH);
        DfeSoftOut so{};
        so.use_threshold = use_threshold ? 1 : 0; so.threshold = (float)threshold;
        so.hFull = hImg; so.wFull = wImg; so.full = full; so.full_conf = full_conf; so.index = (long long *)index; so.scores = scores;
        bool done = false;
        rc = dfe_feat_matching_flat_soft(ctx, in1, pitch1, plane1, in2, K, H1, W1, maxh, maxw, &so, &done);
        if (rc || done) return rc;
        return dfe_fail(ctx, DFE_E_UNSUPPORTED, "dfe_flow_pair_filtered_f32: the matcher declined a shape its predicate took");
    }
    if (!nlayers) {
        hipLaunchKernelGGL(ss_crop_kernel, dim3(ss_grid((long long)K * P1)), dim3(256), 0, ctx->stream, I0, (long long)H * W, W, ny, nx, H1, W1, (long long)K * P1, in1c);
        DFE_LAUNCH_CHECK(ctx);
        in1 = in1c;
    }
    {
        DfeStageScope st(ctx, DFE_STAGE_MATCH);
        rc = dfe_spatial_matching_dispatch(ctx, in1, in2, K, H1, W1, maxh, maxw, vol);
        if (rc) return rc;
        rc = dfe_softmin_f32(ctx, vol, P1, N, prob);                // nn.Minus + the soft-max over the window
        if (rc) return rc;
    }
    DfeStageScope st(ctx, DFE_STAGE_EXTRACT);
    long long *idx_d = index ? (long long *)index : idx_s;
    float *sc_d = scores ? scores : sc_s;
    const int middle = (maxw + 1) / 2 + maxw * ((maxh + 1) / 2 - 1);
    if (!use_threshold) {
        rc = dfe_argbest_center(ctx, prob, P1, N, middle, 1, (int64_t *)idx_d, nullptr);
        if (rc) return rc;
        if (scores) DFE_HIP(ctx, hipMemsetAsync(scores, 0, (size_t)P1 * sizeof(float), ctx->stream));
    } else {
        hipLaunchKernelGGL(ss_fill_kernel, dim3(ss_grid(P1)), dim3(256), 0, ctx->stream, idx_d, sc_d, P1, (long long)middle);
        DFE_LAUNCH_CHECK(ctx);
        rc = dfe_extract_output(ctx, prob, H1, W1, N, sc_d, 0.11, (int64_t *)idx_d);
        if (rc) return rc;
    }
    if (full || full_conf) {
        hipLaunchKernelGGL(ss_paste_kernel, dim3(ss_grid(P1)), dim3(256), 0, ctx->stream, idx_d, sc_d, H1, W1, maxh, maxw, use_threshold ? 1 : 0, (float)threshold, ho, wo,
                           wImg, (long long)hImg * wImg, full, full_conf);
        DFE_LAUNCH_CHECK(ctx);
    }
    return DFE_OK;
}

// nn.SpatialMatching whose first input is a VIEW (rows pitch1 floats apart, planes plane1 floats apart): prepareInput's narrow handed on
// without the contiguous copy
extern "C" int dfe_spatial_matching_strided_f32(dfe_ctx *ctx, const float *in1, int in1_pitch, int64_t in1_plane, const float *in2, int K, int H1, int W1, int maxh,
                                                int maxw, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, in1 && in2 && out, DFE_E_ARG, "dfe_spatial_matching_strided_f32: NULL tensor");
    DFE_REQUIRE(ctx, K > 0 && H1 > 0 && W1 > 0 && maxh > 0 && maxw > 0, DFE_E_SHAPE, "dfe_spatial_matching_strided_f32: K=%d H1=%d W1=%d maxh=%d maxw=%d must be positive", K, H1,
                W1, maxh, maxw);
    DFE_REQUIRE(ctx, in1_pitch >= W1 && in1_plane >= (int64_t)(H1 - 1) * in1_pitch + W1, DFE_E_SHAPE, "dfe_spatial_matching_strided_f32: pitch %d / plane %lld too small for %dx%d",
                in1_pitch, (long long)in1_plane, H1, W1);
    if (in1_pitch == W1 && in1_plane == (int64_t)H1 * W1) return dfe_spatial_matching_dispatch(ctx, in1, in2, K, H1, W1, maxh, maxw, out);
    bool done = false;
    int rc = dfe_feat_matching_flat_strided(ctx, in1, in1_pitch, in1_plane, in2, K, H1, W1, maxh, maxw, out, &done);
    if (rc || done) return rc;
    void *scr = nullptr;
    rc = dfe_scratch(ctx, (size_t)K * H1 * W1 * sizeof(float), &scr);
    if (rc) return rc;
    // (a view with rows pitch apart inside planes plane apart: the crop kernel's (plane, W) pair)
    hipLaunchKernelGGL(ss_crop_kernel, dim3(ss_grid((long long)K * H1 * W1)), dim3(256), 0, ctx->stream, in1, (long long)in1_plane, in1_pitch, 0, 0, H1, W1,
                       (long long)K * H1 * W1, (float *)scr);
    DFE_LAUNCH_CHECK(ctx);
    return dfe_spatial_matching_dispatch(ctx, (const float *)scr, in2, K, H1, W1, maxh, maxw, out);
}
