// filters.hip -- A15 / next-row N1: the learned patch-feature stack in front of the matcher (getFilter,
// opticalflow_model.lua:45-79, radial/radial_opticalflow_network.lua:6-30): nn.SpatialConvolution, nn.SpatialConvolutionMap,
// nn.Tanh.  Direct form, one thread per output element, the accumulation order of the CPU restatement (input plane or
// connection, then ky, kx) so results are bit-identical to it.  Untuned: this is the one place on the path where an
// implicit-GEMM MFMA kernel applies; the layers are a few percent of a matcher pass at the reference's sizes.
#include "dfe_internal.h"

namespace {

__global__ void conv_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias, int nIn, int nOut,
                            int H, int W, int kH, int kW, float *__restrict__ out) {
#pragma clang fp contract(off)
    const int Ho = H - kH + 1, Wo = W - kW + 1;
    const long long n = (long long)nOut * Ho * Wo;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(e % Wo);
        const long long t = e / Wo;
        const int y = (int)(t % Ho), o = (int)(t / Ho);
        float s = bias ? bias[o] : 0.f;
        for (int i = 0; i < nIn; ++i)
            for (int u = 0; u < kH; ++u)
                for (int v = 0; v < kW; ++v)
                    s = s + w[(((long long)o * nIn + i) * kH + u) * kW + v] * in[((long long)i * H + y + u) * W + x + v];
        out[e] = s;
    }
}

__global__ void conv_map_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
                                const int *__restrict__ conn, int nConn, int nOut, int H, int W, int kH, int kW, float *__restrict__ out) {
#pragma clang fp contract(off)
    const int Ho = H - kH + 1, Wo = W - kW + 1;
    const long long n = (long long)nOut * Ho * Wo;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(e % Wo);
        const long long t = e / Wo;
        const int y = (int)(t % Ho), o = (int)(t / Ho);
        float s = bias ? bias[o] : 0.f;
        for (int c = 0; c < nConn; ++c) {
            if (conn[2 * c + 1] - 1 != o) continue;
            const int i = conn[2 * c] - 1;
            for (int u = 0; u < kH; ++u)
                for (int v = 0; v < kW; ++v) s = s + w[((long long)c * kH + u) * kW + v] * in[((long long)i * H + y + u) * W + x + v];
        }
        out[e] = s;
    }
}

// the same two kernels with nn.Tanh applied to the result (one launch per layer of a filter stack inside the one-call pipelines)
template <bool TANH>
__global__ void conv_layer_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias, const int *__restrict__ conn,
                                  int nConn, int nIn, int nOut, int H, int W, int kH, int kW, float *__restrict__ out) {
#pragma clang fp contract(off)
    const int Ho = H - kH + 1, Wo = W - kW + 1;
    const long long n = (long long)nOut * Ho * Wo;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(e % Wo);
        const long long t = e / Wo;
        const int y = (int)(t % Ho), o = (int)(t / Ho);
        float s = bias ? bias[o] : 0.f;
        if (conn) {
            for (int c = 0; c < nConn; ++c) {
                if (conn[2 * c + 1] - 1 != o) continue;
                const int i = conn[2 * c] - 1;
                for (int u = 0; u < kH; ++u)
                    for (int v = 0; v < kW; ++v) s = s + w[((long long)c * kH + u) * kW + v] * in[((long long)i * H + y + u) * W + x + v];
            }
        } else {
            for (int i = 0; i < nIn; ++i)
                for (int u = 0; u < kH; ++u)
                    for (int v = 0; v < kW; ++v)
                        s = s + w[(((long long)o * nIn + i) * kH + u) * kW + v] * in[((long long)i * H + y + u) * W + x + v];
        }
        out[e] = TANH ? tanhf(s) : s;
    }
}

__global__ void tanh_kernel(const float *__restrict__ in, long long n, float *__restrict__ out) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) out[e] = tanhf(in[e]);
}

int grid_n(long long n) {
    long long b = (n + 255) / 256;
    if (b > 256 * 32) b = 256 * 32;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

int dfe_filter_layer_forward(dfe_ctx *ctx, const float *in, const dfe_filter_layer &L, int H, int W, float *out) {
    DFE_REQUIRE(ctx, in && L.weight && out, DFE_E_ARG, "filter layer: NULL tensor");
    DFE_REQUIRE(ctx, L.nIn > 0 && L.nOut > 0 && L.kH > 0 && L.kW > 0 && H >= L.kH && W >= L.kW && (!L.conn || L.nConn > 0), DFE_E_SHAPE,
                "filter layer: %d->%d planes, %dx%d kernel on %dx%d", L.nIn, L.nOut, L.kH, L.kW, H, W);
    const int g = grid_n((long long)L.nOut * (H - L.kH + 1) * (W - L.kW + 1));
    if (L.tanh_after)
        hipLaunchKernelGGL(conv_layer_kernel<true>, dim3(g), dim3(256), 0, ctx->stream, in, L.weight, L.bias, (const int *)L.conn, L.nConn, L.nIn, L.nOut, H, W, L.kH, L.kW, out);
    else
        hipLaunchKernelGGL(conv_layer_kernel<false>, dim3(g), dim3(256), 0, ctx->stream, in, L.weight, L.bias, (const int *)L.conn, L.nConn, L.nIn, L.nOut, H, W, L.kH, L.kW, out);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

extern "C" {

int dfe_spatial_convolution_f32(dfe_ctx *ctx, const float *in, const float *weight, const float *bias, int nIn, int nOut, int H, int W,
                                int kH, int kW, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, in && weight && out, DFE_E_ARG, "dfe_spatial_convolution_f32: NULL tensor");
    DFE_REQUIRE(ctx, nIn > 0 && nOut > 0 && kH > 0 && kW > 0 && H >= kH && W >= kW, DFE_E_SHAPE,
                "dfe_spatial_convolution_f32: %d->%d planes, %dx%d kernel on %dx%d", nIn, nOut, kH, kW, H, W);
    hipLaunchKernelGGL(conv_kernel, dim3(grid_n((long long)nOut * (H - kH + 1) * (W - kW + 1))), dim3(256), 0, ctx->stream, in, weight, bias,
                       nIn, nOut, H, W, kH, kW, out);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_spatial_convolution_map_f32(dfe_ctx *ctx, const float *in, const float *weight, const float *bias, const int32_t *conn, int nConn,
                                    int nIn, int nOut, int H, int W, int kH, int kW, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, in && weight && conn && out, DFE_E_ARG, "dfe_spatial_convolution_map_f32: NULL tensor");
    DFE_REQUIRE(ctx, nIn > 0 && nOut > 0 && nConn > 0 && kH > 0 && kW > 0 && H >= kH && W >= kW, DFE_E_SHAPE,
                "dfe_spatial_convolution_map_f32: %d connections %d->%d planes, %dx%d kernel on %dx%d", nConn, nIn, nOut, kH, kW, H, W);
    hipLaunchKernelGGL(conv_map_kernel, dim3(grid_n((long long)nOut * (H - kH + 1) * (W - kW + 1))), dim3(256), 0, ctx->stream, in, weight,
                       bias, conn, nConn, nOut, H, W, kH, kW, out);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_tanh_f32(dfe_ctx *ctx, const float *in, int64_t n, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, n >= 0, DFE_E_SHAPE, "dfe_tanh_f32: n=%lld", (long long)n);
    if (n == 0) return DFE_OK;
    DFE_REQUIRE(ctx, in && out, DFE_E_ARG, "dfe_tanh_f32: NULL tensor");
    hipLaunchKernelGGL(tanh_kernel, dim3(grid_n(n)), dim3(256), 0, ctx->stream, in, (long long)n, out);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// nn.SpatialContrastiveNormalization(nIn, 1-D kernel, threshold, thresval): version2/network.lua:12 puts it in front of both
// filter branches (image.gaussian1D(normalization_k)).  = SpatialSubtractiveNormalization then SpatialDivisiveNormalization,
// each built on one "estimator": zero pad, horizontal pass per plane, vertical pass that also sums the planes, divided by the
// estimator of a tensor of ones (border correction).  Un-vendored nn, nothing in the reference tests it: restated from recall
// (the CPU restatement states the same recall) -- parity unpinned.  Same term order as that restatement: bit-identical.
// ------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int CN_MAXK = 33;
struct CnKernel { float kn[CN_MAXK]; int k; };

// mode 0: src = in; 1: src = (in - est/coef)^2 (the divisive stage's input, recomputed instead of stored)
template <int MODE>
__global__ void cn_rows_kernel(const float *__restrict__ in, const float *__restrict__ est, const float *__restrict__ coef, int C, int H, int W,
                               CnKernel kk, int ones, float *__restrict__ tmp) {
#pragma clang fp contract(off)
    const long long n = (long long)C * H * W, P = (long long)H * W;
    const int pl = kk.k / 2;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(e % W);
        const long long row = e - x;
        const long long p0 = (e % P) - x;
        float s = 0.f;
        for (int v = 0; v < kk.k; ++v) {
            const int xx = x + v - pl;
            float a = 0.f;
            if (xx >= 0 && xx < W) {
                if (ones) a = 1.f;
                else if (MODE == 0) a = in[row + xx];
                else { const float y = in[row + xx] - est[p0 + xx] / coef[p0 + xx]; a = y * y; }
            }
            s = s + kk.kn[v] * a;
        }
        tmp[e] = s;
    }
}

__global__ void cn_cols_kernel(const float *__restrict__ tmp, int C, int H, int W, CnKernel kk, float *__restrict__ out) {
#pragma clang fp contract(off)
    const long long P = (long long)H * W;
    const int pl = kk.k / 2;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < P; e += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(e % W), y = (int)(e / W);
        float s = 0.f;
        for (int c = 0; c < C; ++c)
            for (int u = 0; u < kk.k; ++u) {
                const int yy = y + u - pl;
                const float a = (yy >= 0 && yy < H) ? tmp[c * P + (long long)yy * W + x] : 0.f;
                s = s + kk.kn[u] * a;
            }
        out[e] = s;
    }
}

__global__ void cn_finish_kernel(const float *__restrict__ in, const float *__restrict__ est, const float *__restrict__ est2,
                                 const float *__restrict__ coef, int C, long long P, float threshold, float thresval, float *__restrict__ out) {
#pragma clang fp contract(off)
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < C * P; e += (long long)gridDim.x * blockDim.x) {
        const long long p = e % P;
        const float y = in[e] - est[p] / coef[p];
        float sd = sqrtf(est2[p]) / coef[p];
        sd = sd > threshold ? sd : thresval;
        out[e] = y / sd;
    }
}

}  // namespace

extern "C" int dfe_contrastive_normalization_f32(dfe_ctx *ctx, const float *in, int C, int H, int W, const float *kernel_host, int k, float threshold,
                                                 float thresval, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, in && kernel_host && out, DFE_E_ARG, "dfe_contrastive_normalization_f32: NULL argument");
    DFE_REQUIRE(ctx, C > 0 && H > 0 && W > 0 && k > 0 && k <= CN_MAXK, DFE_E_SHAPE, "dfe_contrastive_normalization_f32: C=%d %dx%d kernel %d (max %d)", C, H,
                W, k, CN_MAXK);
    CnKernel kk;
    kk.k = k;
    float ks = 0.f;
    for (int i = 0; i < k; ++i) ks += kernel_host[i];
    for (int i = 0; i < k; ++i) kk.kn[i] = kernel_host[i] / (ks * (float)C);       // self.kernel:div(self.kernel:sum() * self.nInputPlane)
    const long long P = (long long)H * W;
    void *scr = nullptr;
    int rc = dfe_scratch(ctx, ((size_t)C * P + 3 * P) * sizeof(float), &scr);
    if (rc) return rc;
    float *tmp = (float *)scr, *coef = tmp + C * P, *est = coef + P, *est2 = est + P;
    const int g1 = grid_n(C * P), g2 = grid_n(P);
    hipLaunchKernelGGL(cn_rows_kernel<0>, dim3(g1), dim3(256), 0, ctx->stream, in, (const float *)nullptr, (const float *)nullptr, C, H, W, kk, 1, tmp);
    hipLaunchKernelGGL(cn_cols_kernel, dim3(g2), dim3(256), 0, ctx->stream, tmp, C, H, W, kk, coef);
    hipLaunchKernelGGL(cn_rows_kernel<0>, dim3(g1), dim3(256), 0, ctx->stream, in, (const float *)nullptr, (const float *)nullptr, C, H, W, kk, 0, tmp);
    hipLaunchKernelGGL(cn_cols_kernel, dim3(g2), dim3(256), 0, ctx->stream, tmp, C, H, W, kk, est);
    hipLaunchKernelGGL(cn_rows_kernel<1>, dim3(g1), dim3(256), 0, ctx->stream, in, est, coef, C, H, W, kk, 0, tmp);
    hipLaunchKernelGGL(cn_cols_kernel, dim3(g2), dim3(256), 0, ctx->stream, tmp, C, H, W, kk, est2);
    hipLaunchKernelGGL(cn_finish_kernel, dim3(g1), dim3(256), 0, ctx->stream, in, est, est2, coef, C, P, threshold, thresval, out);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}
