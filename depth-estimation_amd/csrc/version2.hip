// version2.hip -- the single-scale learned model of version2/ as ONE call: what version2/test.lua:40-53 does with getNetwork(datap)
// (version2/network.lua:5-39) for a frame pair:
//   filter1 = SpatialContrastiveNormalization(C, gaussian1D(normalization_k)) -> SpatialPadding(-lWin, -tWin, -rWin, -bWin) -> conv stack
//   filter2 = the same normalisation -> the same conv stack (shared weights)
//   SpatialMatching(hWin, wWin) on the two feature maps, then the dense decode of test.lua:45-51: first minimum over the window,
//   yflow = floor(idx / wWin) - tWin, xflow = idx - floor(idx / wWin) * wWin - lWin.
// Every stage is the stand-alone device op (same kernels, same order of operations), so the one-call result equals the staged host
// path bit for bit.
#include "dfe_internal.h"

namespace {

// one wave per pixel: first minimum of the window's N cells (index order), decoded as version2/test.lua:45-51 does
__global__ __launch_bounds__(256) void v2_argmin_decode_kernel(const float *__restrict__ vol, long long P, int N, int wWin, int lWin, int tWin,
                                                               long long *__restrict__ idx, float *__restrict__ xflow, float *__restrict__ yflow) {
    const int lane = threadIdx.x & 63;
    const long long wid = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long px = wid; px < P; px += nw) {
        const float *v = vol + px * N;
        float best = __int_as_float(0x7f800000);
        int bi = 0x7fffffff;
        for (int i = lane; i < N; i += 64) {
            const float c = v[i];
            if (c < best) { best = c; bi = i; }       // (strict: the lane keeps its first minimum; a NaN never wins -- torch.min would return it: the
                                                      //  equality with the staged host path holds for volumes without NaN)
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) {
            if (bi == 0x7fffffff) bi = 0;
            const int fy = bi / wWin;
            if (idx) idx[px] = (long long)bi + 1;
            if (yflow) yflow[px] = (float)(fy - tWin);
            if (xflow) xflow[px] = (float)(bi - fy * wWin - lWin);
        }
    }
}

int v2_grid(long long n, int per_block) {
    long long b = (n + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : b > 65535 * 16 ? 65535 * 16 : b);
}

}  // namespace

extern "C" int dfe_version2_flow_pair_f32(dfe_ctx *ctx, const float *prev, const float *cur, int C, int H, int W, const float *norm_kernel_host,
                                          int norm_k, float threshold, float thresval, const dfe_filter_layer *layers, int nlayers, int hWin,
                                          int wWin, float *xflow, float *yflow, int64_t *idx, float *volume) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, prev && cur && norm_kernel_host && layers, DFE_E_ARG, "dfe_version2_flow_pair_f32: NULL argument");
    DFE_REQUIRE(ctx, C > 0 && H > 0 && W > 0 && nlayers > 0 && nlayers <= 8 && hWin > 0 && wWin > 0, DFE_E_ARG,
                "dfe_version2_flow_pair_f32: C=%d %dx%d, %d layers, window %dx%d", C, H, W, nlayers, hWin, wWin);
    DFE_REQUIRE(ctx, layers[0].nIn == C, DFE_E_SHAPE, "dfe_version2_flow_pair_f32: the first layer reads %d planes, the frames have %d", layers[0].nIn, C);
    // datap.lWin = ceil(wWin / 2) - 1, tWin = ceil(hWin / 2) - 1, rWin = floor(wWin / 2), bWin = floor(hWin / 2) (version2/test.lua:18-21)
    const int lWin = (wWin + 1) / 2 - 1, tWin = (hWin + 1) / 2 - 1;
    const int Hc = H - (hWin - 1), Wc = W - (wWin - 1);           // the cropped branch's frame
    int hk = 1, wk = 1;
    for (int i = 0; i < nlayers; ++i) {
        DFE_REQUIRE(ctx, layers[i].weight && layers[i].kH > 0 && layers[i].kW > 0 && layers[i].nIn > 0 && layers[i].nOut > 0, DFE_E_ARG,
                    "dfe_version2_flow_pair_f32: layer %d is incomplete", i);
        DFE_REQUIRE(ctx, i == 0 || layers[i].conn || layers[i].nIn == layers[i - 1].nOut, DFE_E_SHAPE, "dfe_version2_flow_pair_f32: layer %d reads %d planes, layer %d makes %d", i,
                    layers[i].nIn, i - 1, layers[i - 1].nOut);
        hk += layers[i].kH - 1;
        wk += layers[i].kW - 1;
    }
    const int H1 = Hc - hk + 1, W1 = Wc - wk + 1;                 // matcher output = the dense inference region
    DFE_REQUIRE(ctx, H1 > 0 && W1 > 0, DFE_E_SHAPE, "dfe_version2_flow_pair_f32: frame %dx%d too small for window %dx%d + kernel %dx%d", H, W, hWin, wWin, hk, wk);
    const int K = layers[nlayers - 1].nOut, N = hWin * wWin;
    const long long P = (long long)H * W, P1 = (long long)H1 * W1;
    // arena: normalisation scratch | normalised cur | cropped normalised prev | two ping-pong feature buffers per branch | volume (unless the caller gave one)
    int maxplanes = C;
    for (int i = 0; i < nlayers; ++i) maxplanes = maxplanes > layers[i].nOut ? maxplanes : layers[i].nOut;
    const size_t f_cn = ((size_t)C + 3) * P, f_n = (size_t)C * P, f_c = (size_t)C * Hc * Wc;
    const size_t f_fa = (size_t)maxplanes * Hc * Wc, f_fb = (size_t)maxplanes * P;
    // (the volume needs a place in the arena only when nobody gave one AND the matcher + arg-min kernel will not take the shape:
    //  316 MB at VGA, 2.4 GB at 1080p that the lean path never touches)
    const bool lean = !volume && (xflow || yflow || idx) && dfe_feat_matching_flat_argmin_takes(ctx, K, H1, W1, hWin, wWin);
    const size_t f_vol = (volume || lean) ? 0 : (size_t)P1 * N;
    auto al = [](size_t f) { return (f + 63) / 64 * 64; };
    // both matrix-core options on: the last layer's convolution leaves the features' squared norms for the matcher (no pass of its own)
    const bool mm_both = ctx->opt[DFE_OPT_CONV_MFMA] > 0 && dfe_feat_matching_mfma_takes(ctx, K, H1, W1, hWin, wWin) && layers[nlayers - 1].nOut <= 32 &&
                         !layers[nlayers - 1].conn;
    const size_t f_nrm = mm_both ? dfe_feat_matching_mfma_scratch(H1, W1, hWin, wWin) : 0;
    void *scr = nullptr;
    int rc = dfe_scratch(ctx, (al(f_cn) + al(f_n) + al(f_c) + 2 * al(f_fa) + 2 * al(f_fb) + al(f_vol) + al(f_nrm)) * sizeof(float), &scr);
    if (rc) return rc;
    float *s_cn = (float *)scr, *n1 = s_cn + al(f_cn), *c0 = n1 + al(f_n);
    float *fa[2] = {c0 + al(f_c), c0 + al(f_c) + al(f_fa)};
    float *fb[2] = {fa[1] + al(f_fa), fa[1] + al(f_fa) + al(f_fb)};
    float *vol = volume ? volume : fb[1] + al(f_fb);
    float *nrm = fb[1] + al(f_fb) + al(f_vol);                    // |a|^2 [H1][W1] | |b|^2 [H2][W2]
    bool norms_ready = false;
    const float *ia = c0, *ib = n1;
    {
        DfeStageScope st(ctx, DFE_STAGE_FILTER);
        // both frames in the normalisation's two launches; the first branch's crop (SpatialPadding(-lWin, -tWin, -rWin, -bWin)) is the
        // window its last launch writes
        rc = dfe_contrastive_normalization_run2(ctx, prev, cur, C, H, W, norm_kernel_host, norm_k, threshold, thresval, s_cn, c0, n1, lWin, tWin, Wc, Hc);
        if (rc) return rc;
        int ha = Hc, wa = Wc, hb = H, wb = W;
        for (int i = 0; i < nlayers; ++i) {
            const float *in2[2] = {ia, ib};
            const dfe_filter_layer *L2[2] = {&layers[i], &layers[i]};
            const int H2[2] = {ha, hb}, W2[2] = {wa, wb};
            float *o2[2] = {fa[i & 1], fb[i & 1]};
            bool mm = false;
            if (ctx->opt[DFE_OPT_CONV_MFMA] > 0) {   // opt-in: the layer as an implicit GEMM on the matrix cores (fused multiply-adds)
                float *n2[2] = {nrm, nrm + (size_t)H1 * W1};
                const bool last = mm_both && i == nlayers - 1;
                rc = dfe_conv_mfma_res_batch(ctx, 2, in2, H2, W2, nullptr, nullptr, layers[i], o2, &mm, last ? n2 : nullptr);
                if (rc) return rc;
                norms_ready = mm && last;
            }
            if (!mm) rc = dfe_filter_layer_forward_batch(ctx, 2, in2, L2, H2, W2, o2);
            if (rc) return rc;
            ia = o2[0]; ib = o2[1];
            ha -= layers[i].kH - 1; wa -= layers[i].kW - 1; hb -= layers[i].kH - 1; wb -= layers[i].kW - 1;
        }
        DFE_REQUIRE(ctx, ha == H1 && wa == W1 && hb == H1 + hWin - 1 && wb == W1 + wWin - 1, DFE_E_SHAPE, "dfe_version2_flow_pair_f32: internal shape mismatch");
    }
    if (!volume && (xflow || yflow || idx)) {
        // nobody asked for the volume: matcher and decode in one kernel where the flat-tile matcher takes the shape (bit-identical to the
        // volume path: the same sums, the same first minimum)
        DfeStageScope st(ctx, DFE_STAGE_MATCH);
        bool done = false;
        if (norms_ready) {
            rc = dfe_feat_matching_mfma(ctx, ia, ib, K, H1, W1, hWin, wWin, nrm, nullptr, (long long *)idx, xflow, yflow, &done, true);
            if (rc || done) return rc;
        }
        rc = dfe_feat_matching_flat_argmin(ctx, ia, ib, K, H1, W1, hWin, wWin, (long long *)idx, xflow, yflow, &done);
        if (rc || done) return rc;
        DFE_REQUIRE(ctx, !lean, DFE_E_UNSUPPORTED, "dfe_version2_flow_pair_f32: the matcher declined a shape its predicate took");
    }
    {
        DfeStageScope st(ctx, DFE_STAGE_MATCH);
        rc = dfe_spatial_matching_dispatch(ctx, ia, ib, K, H1, W1, hWin, wWin, vol);
        if (rc) return rc;
    }
    {
        DfeStageScope st(ctx, DFE_STAGE_EXTRACT);
        if (xflow || yflow || idx) {
            hipLaunchKernelGGL(v2_argmin_decode_kernel, dim3(v2_grid(P1, 4)), dim3(256), 0, ctx->stream, vol, P1, N, wWin, lWin, tWin, (long long *)idx, xflow, yflow);
            DFE_LAUNCH_CHECK(ctx);
        }
    }
    return DFE_OK;
}

// nn.SpatialMatching(maxh, maxw) followed by `output:min(3)` and the decode of version2/test.lua:45-51 (radial_opticalflow_groundtruth.lua:88-100
// without the centre override), on feature maps that are already there: matcher and first-minimum decode in one kernel where the flat-tile
// matcher takes the shape, the matcher into the scratch arena + the decode kernel otherwise -- the same sums and the same first minimum
// either way.
extern "C" int dfe_spatial_matching_argmin_f32(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh, int maxw,
                                               int64_t *idx, float *xflow, float *yflow) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, in1 && in2 && (idx || xflow || yflow), DFE_E_ARG, "dfe_spatial_matching_argmin_f32: NULL argument");
    DFE_REQUIRE(ctx, K > 0 && H1 > 0 && W1 > 0 && maxh > 0 && maxw > 0, DFE_E_ARG, "dfe_spatial_matching_argmin_f32: K=%d %dx%d window %dx%d", K, H1, W1, maxh, maxw);
    const int lWin = (maxw + 1) / 2 - 1, tWin = (maxh + 1) / 2 - 1;
    bool done = false;
    {
        DfeStageScope st(ctx, DFE_STAGE_MATCH);
        int rc = dfe_feat_matching_flat_argmin(ctx, in1, in2, K, H1, W1, maxh, maxw, (long long *)idx, xflow, yflow, &done);
        if (rc || done) return rc;
    }
    const long long P1 = (long long)H1 * W1;
    const int N = maxh * maxw;
    void *scr = nullptr;
    int rc = dfe_scratch(ctx, (size_t)P1 * N * sizeof(float), &scr);
    if (rc) return rc;
    {
        DfeStageScope st(ctx, DFE_STAGE_MATCH);
        rc = dfe_spatial_matching_dispatch(ctx, in1, in2, K, H1, W1, maxh, maxw, (float *)scr);
        if (rc) return rc;
    }
    DfeStageScope st(ctx, DFE_STAGE_EXTRACT);
    hipLaunchKernelGGL(v2_argmin_decode_kernel, dim3(v2_grid(P1, 4)), dim3(256), 0, ctx->stream, (const float *)scr, P1, N, maxw, lWin, tWin, (long long *)idx, xflow, yflow);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}
