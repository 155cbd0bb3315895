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
