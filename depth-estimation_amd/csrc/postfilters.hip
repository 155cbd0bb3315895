// postfilters.hip -- A16 postProcessImage (masked mode / median flow filters), A17 enlargeMask,
// A18 nn.OutputExtractor (soft arg-max expectation).
//   replaces: postProcessImage inline C `fmax` / `fmed`   opticalflow_model.lua:323-472
//             enlargeMask inline C                        depth_estimation_api.lua:76-132
//             nn.OutputExtractor:updateOutput              OutputExtractor.lua:21-35
#include "dfe_internal.h"
#include <cmath>

namespace {

int grid1d(long long n, int per = 256) {
    long long b = (n + per - 1) / per;
    if (b > 256 * 32) b = 256 * 32;
    if (b < 1) b = 1;
    return (int)b;
}

// floor(flow + 0.5) and the global min / max of the result (over both planes) through ordered-int atomics
__device__ __forceinline__ int ordered(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__global__ void round_minmax_kernel(const float *__restrict__ flow, long long n, float *__restrict__ R, int *__restrict__ mm) {
    int lo = 0x7fffffff, hi = (int)0x80000000;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        float r = floorf(flow[e] + 0.5f);   // (input+0.5):floor()  opticalflow_model.lua:437
        R[e] = r;
        int o = ordered(r);
        lo = min(lo, o); hi = max(hi, o);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { lo = min(lo, __shfl_xor(lo, off)); hi = max(hi, __shfl_xor(hi, off)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&mm[0], lo); atomicMax(&mm[1], hi); }
}
__device__ __forceinline__ float unordered(int o) { return __int_as_float(o >= 0 ? o : o ^ 0x7fffffff); }

// fmax: per window the most frequent (vx + 16*vy) among masked pixels, lowest code on ties (strict '>' scan
// from code 0, :373-377); empty window -> code 0.  Counted by comparing window elements pairwise instead of a
// 256-bin histogram per thread.
__global__ void mode_filter_kernel(const float *__restrict__ R, const float *__restrict__ mask, int H, int W, int k,
                                   const int *__restrict__ mm, float *__restrict__ out) {
    const long long HW = (long long)H * W;
    const float m = unordered(mm[0]);
    const int halfk = k / 2;
    const int nh = H - k, nw = W - k;   // loop bounds `< h-k`, `< w-k` as shipped (:356-357)
    const long long total = (long long)(nh > 0 ? nh : 0) * (nw > 0 ? nw : 0);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e / nw), j = (int)(e - (long long)i * nw);
        int best_code = 0, best_cnt = 0;
        for (int a = 0; a < k * k; ++a) {
            const int ia = i + a / k, ja = j + a % k;
            if (!mask[(long long)ia * W + ja]) continue;
            const int ca = (int)(R[HW + (long long)ia * W + ja] - m) + 16 * (int)(R[(long long)ia * W + ja] - m);
            int cnt = 0;
            for (int b = 0; b < k * k; ++b) {
                const int ib = i + b / k, jb = j + b % k;
                if (!mask[(long long)ib * W + jb]) continue;
                const int cb = (int)(R[HW + (long long)ib * W + jb] - m) + 16 * (int)(R[(long long)ib * W + jb] - m);
                cnt += (cb == ca);
            }
            if (cnt > best_cnt || (cnt == best_cnt && ca < best_code)) { best_cnt = cnt; best_code = ca; }
        }
        if (best_cnt == 0) best_code = 0;
        out[HW + (long long)(i + halfk) * W + j + halfk] = (float)(best_code % 16);
        out[(long long)(i + halfk) * W + j + halfk] = (float)(best_code / 16);
    }
}
__global__ void add_scalar_kernel(float *__restrict__ out, long long n, const int *__restrict__ mm) {
    const float m = unordered(mm[0]);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) out[e] += m;   // output+m :440
}

// fmed: per-component median tmp[n/2] of the masked window values; empty window -> 0 (zeroed buffers :413-414)
__global__ void median_filter_kernel(const float *__restrict__ flow, const float *__restrict__ mask, int H, int W, int k,
                                     float *__restrict__ out) {
    const long long HW = (long long)H * W;
    const int halfk = k / 2, nh = H - k, nw = W - k;
    const long long total = (long long)(nh > 0 ? nh : 0) * (nw > 0 ? nw : 0);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e / nw), j = (int)(e - (long long)i * nw);
        float med[2];
        for (int pl = 0; pl < 2; ++pl) {
            float t[32];
            int n = 0;
            for (int a = 0; a < k * k; ++a) {
                const int ia = i + a / k, ja = j + a % k;
                if (mask[(long long)ia * W + ja]) {
                    float v = flow[pl * HW + (long long)ia * W + ja];
                    int q = n++;
                    while (q > 0 && t[q - 1] > v) { t[q] = t[q - 1]; --q; }   // insertion sort == qsort order for floats
                    t[q] = v;
                }
            }
            med[pl] = n ? t[n / 2] : 0.f;
        }
        out[(long long)(i + halfk) * W + j + halfk] = med[0];
        out[HW + (long long)(i + halfk) * W + j + halfk] = med[1];
    }
}

// enlargeMask: rows first, then columns on the row-eroded mask (depth_estimation_api.lua:93-126)
__global__ void enlarge_rows_kernel(float *__restrict__ mask, int H, int W, int ix) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H; i += gridDim.x * blockDim.x) {
        float *row = mask + (long long)i * W;
        for (int j = 0; j < W; ++j)
            if (row[j] > 0.5f) { for (int k = j; k < min(j + ix, W); ++k) row[k] = 0.f; break; }
        for (int j = W - 1; j >= 0; --j)
            if (row[j] > 0.5f) { for (int k = j; k >= max(j - ix + 1, 0); --k) row[k] = 0.f; break; }
    }
}
__global__ void enlarge_cols_kernel(float *__restrict__ mask, int H, int W, int iy) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < W; j += gridDim.x * blockDim.x) {
        for (int i = 0; i < H; ++i)
            if (mask[(long long)i * W + j] > 0.5f) { for (int k = i; k < min(i + iy, H); ++k) mask[(long long)k * W + j] = 0.f; break; }
        for (int i = H - 1; i >= 0; --i)
            if (mask[(long long)i * W + j] > 0.5f) { for (int k = i; k >= max(i - iy + 1, 0); --k) mask[(long long)k * W + j] = 0.f; break; }
    }
}

// OutputExtractor: x = sum_k p_k * j(k), y = sum_k p_k * i(k), 1-based cell coordinates; one wave per pixel,
// sequential-order partial sums per lane then a butterfly (tolerance documented in the tests)
__global__ __launch_bounds__(256) void output_extractor_kernel(const float *__restrict__ in, long long P, int maxh, int maxw,
                                                               float *__restrict__ x, float *__restrict__ y) {
    const int lane = threadIdx.x & 63, N = maxh * maxw;
    for (long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); p < P; p += (long long)gridDim.x * 4) {
        float sx = 0.f, sy = 0.f;
        for (int k = lane; k < N; k += 64) {
            float v = in[p * N + k];
            sx += v * (float)(k % maxw + 1);
            sy += v * (float)(k / maxw + 1);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { sx += __shfl_xor(sx, off); sy += __shfl_xor(sy, off); }
        if (lane == 0) { x[p] = sx; y[p] = sy; }
    }
}

}  // namespace

// A11 'mean' extraction: out[p] = sum_b in[p][b], double accumulator (opticalflow_model.lua:192; TH sums floats in double)
__global__ void marginal_sum_kernel(const float *__restrict__ in, long long n, int B, float *__restrict__ out) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        double acc = 0;
        for (int b = 0; b < B; ++b) acc += in[e * B + b];
        out[e] = (float)acc;
    }
}

extern "C" {

int dfe_postprocess_image_f32(dfe_ctx *ctx, const float *flow, const float *mask, int H, int W, int winsize, int method,
                              float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, flow && mask && out && H > 0 && W > 0 && winsize > 0, DFE_E_ARG, "dfe_postprocess_image_f32: bad argument");
    DFE_REQUIRE(ctx, method == 0 || method == 1, DFE_E_ARG, "dfe_postprocess_image_f32: method %d (0 = 'max', 1 = median)", method);
    const long long HW = (long long)H * W;
    DFE_HIP(ctx, hipMemsetAsync(out, 0, 2 * HW * sizeof(float), ctx->stream));   // torch.Tensor(2,h,w):zero() :324
    if (method == 1) {
        DFE_REQUIRE(ctx, winsize * winsize <= 32, DFE_E_ARG,
                    "dfe_postprocess_image_f32: median window %dx%d exceeds the reference's 32-value buffer (opticalflow_model.lua:405)",
                    winsize, winsize);
        hipLaunchKernelGGL(median_filter_kernel, dim3(grid1d(HW)), dim3(256), 0, ctx->stream, flow, mask, H, W, winsize, out);
        DFE_LAUNCH_CHECK(ctx);
        return DFE_OK;
    }
    void *scr = nullptr;
    int rc = dfe_scratch(ctx, 2 * HW * sizeof(float) + 64, &scr);
    if (rc) return rc;
    float *R = (float *)scr;
    int *mm = (int *)((char *)scr + 2 * HW * sizeof(float));
    const int init[2] = {0x7fffffff, (int)0x80000000};
    DFE_HIP(ctx, hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(round_minmax_kernel, dim3(grid1d(2 * HW)), dim3(256), 0, ctx->stream, flow, 2 * HW, R, mm);
    int h[2];
    DFE_HIP(ctx, hipMemcpyAsync(h, mm, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    auto unord = [](int o) { int i = o >= 0 ? o : o ^ 0x7fffffff; float f; memcpy(&f, &i, 4); return f; };
    DFE_REQUIRE(ctx, unord(h[1]) - unord(h[0]) <= 15.f, DFE_E_ARG,
                "dfe_postprocess_image_f32: rounded flow spans %g..%g, more than the reference's 16x16 histogram (opticalflow_model.lua:349-351)",
                unord(h[0]), unord(h[1]));
    hipLaunchKernelGGL(mode_filter_kernel, dim3(grid1d(HW)), dim3(256), 0, ctx->stream, R, mask, H, W, winsize, mm, out);
    hipLaunchKernelGGL(add_scalar_kernel, dim3(grid1d(2 * HW)), dim3(256), 0, ctx->stream, out, 2 * HW, mm);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_enlarge_mask_f32(dfe_ctx *ctx, float *mask, int H, int W, int ix, int iy) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, H >= 0 && W >= 0, DFE_E_SHAPE, "dfe_enlarge_mask_f32: H=%d W=%d", H, W);
    if ((long long)H * W == 0) return DFE_OK;
    DFE_REQUIRE(ctx, mask, DFE_E_ARG, "dfe_enlarge_mask_f32: NULL tensor");
    hipLaunchKernelGGL(enlarge_rows_kernel, dim3(grid1d(H, 64)), dim3(64), 0, ctx->stream, mask, H, W, ix);
    hipLaunchKernelGGL(enlarge_cols_kernel, dim3(grid1d(W, 64)), dim3(64), 0, ctx->stream, mask, H, W, iy);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_output_extractor_f32(dfe_ctx *ctx, const float *input, int64_t P, int maxh, int maxw, float *x, float *y) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, P >= 0 && maxh > 0 && maxw > 0, DFE_E_SHAPE, "dfe_output_extractor_f32: P=%lld window %dx%d", (long long)P, maxh, maxw);
    if (P == 0) return DFE_OK;
    DFE_REQUIRE(ctx, input && x && y, DFE_E_ARG, "dfe_output_extractor_f32: NULL tensor");
    hipLaunchKernelGGL(output_extractor_kernel, dim3(grid1d(P, 4)), dim3(256), 0, ctx->stream, input, (long long)P, maxh, maxw, x, y);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_marginal_sum_f32(dfe_ctx *ctx, const float *in, int64_t P, int A, int B, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, P >= 0 && A > 0 && B > 0, DFE_E_SHAPE, "dfe_marginal_sum_f32: P=%lld A=%d B=%d", (long long)P, A, B);
    if (P == 0) return DFE_OK;
    DFE_REQUIRE(ctx, in && out, DFE_E_ARG, "dfe_marginal_sum_f32: NULL tensor");
    const long long n = (long long)P * A;
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(marginal_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, in, n, B, out);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

}  // extern "C"
