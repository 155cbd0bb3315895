// conv_mfma.hip -- next-row N1: nn.SpatialConvolution as an implicit GEMM on the matrix cores.
//   replaces: the convolutions of getFilter (opticalflow_model.lua:45-79: {3,5,5,4},{4,5,5,4},{4,5,5,10} in
//   tests/time_matching.lua:13), of version2/network.lua:12-19 (3 -> 32 planes, 17 x 17) and of
//   radial/radial_opticalflow_network.lua:6-30, when the caller asks for the fast kernel (dfe_set_convolution_kernel).
// GEMM view: M = output pixels, N = output planes (tiles of 16), K = (input plane, ky, kx).  v_mfma_f32_16x16x4_f32: f32
// operands, f32 accumulate -- on gfx950 bitwise an fmaf chain over k in order (MI355X_MICROARCH.md), so
//     out = fma(w[K-1], in[K-1], ... fma(w[1], in[1], fma(w[0], in[0], bias)))        with k = (i, u, v), v fastest,
// i.e. the reference order with fused instead of separately rounded multiply-adds: tolerance 1e-5 relative to sum |terms|
// against nn.SpatialConvolution's loop, bit-exact against the oracle's fmaf variant.  The matrix cores run f32 at the vector
// FMA rate, but one instruction replaces 4 x 16 x 16 multiply-adds: the direct kernel needs two VALU instructions per
// multiply-add (no fusing: bit-exactness with the CPU loop) and one accumulator register per output.
//   block = 4 waves = 4 output rows x 64 columns; per input plane: the (kH + 3) x (64 + kWp - 1) input tile and that
//   plane's kH x kWp x 16 weights (k-major, zero padded to kWp = 4 ceil(kW/4)) are staged in LDS; per (ky, 4 kx) step a
//   wave reads one B fragment and, for each of its four 16-pixel tiles, one A fragment (overlapping lanes broadcast) + one
//   MFMA.  Zero-padded weights multiply finite tile values (the tile is zero-filled past the frame), so padding adds
//   exact zeros to the chain.
#include "dfe_internal.h"

namespace {

typedef float f4v __attribute__((ext_vector_type(4)));

template <bool TANH>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
                                                       int nIn, int nOut, int H, int W, int kH, int kW, float *__restrict__ out) {
    extern __shared__ float smem[];
    const int Ho = H - kH + 1, Wo = W - kW + 1;
    const int kWp = (kW + 3) & ~3;
    const int TW = 64 + kWp - 1 + 3, TH = kH + 3;      // tile: 4 output rows, 64 columns (+3: the last k-step's lanes read past kWp-1)
    float *tile = smem;                                // [TH][TW]
    float *wl = smem + TH * TW;                        // [kH][kWp][16]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 4, n0 = blockIdx.z * 16;
    const int m = lane & 15, kq = lane >> 4;
    const int n = n0 + m;
    f4v acc[4];
    {
        const float b = (bias && n < nOut) ? bias[n] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f4v{b, b, b, b};
    }
    for (int i = 0; i < nIn; ++i) {
        __syncthreads();
        for (int e = threadIdx.x; e < TH * TW; e += 256) {
            const int r = e / TW, c = e - r * TW;
            const int y = y0 + r, x = x0 + c;
            tile[e] = (y < H && x < W) ? in[((long long)i * H + y) * W + x] : 0.f;
        }
        for (int e = threadIdx.x; e < kH * kWp * 16; e += 256) {
            const int nn = e & 15, v = (e >> 4) % kWp, u = (e >> 4) / kWp;
            wl[e] = (v < kW && n0 + nn < nOut) ? w[(((long long)(n0 + nn) * nIn + i) * kH + u) * kW + v] : 0.f;
        }
        __syncthreads();
        for (int u = 0; u < kH; ++u) {
            const float *trow = tile + (u + wave) * TW + m + kq;
            const float *wrow = wl + (u * kWp + kq) * 16 + m;
            for (int v0 = 0; v0 < kWp; v0 += 4) {
                const float b = wrow[v0 * 16];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(trow[v0 + 16 * t], b, acc[t], 0, 0, 0);
            }
        }
    }
    const int y = y0 + wave;
    if (y < Ho && n < nOut) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int x = x0 + 16 * t + 4 * kq;        // D[m' = 4 kq + r][n = lane & 15]
            float *o = out + ((long long)n * Ho + y) * Wo + x;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (x + r < Wo) o[r] = TANH ? tanhf(acc[t][r]) : acc[t][r];
        }
    }
}

}  // namespace

size_t conv_mfma_lds_bytes(int kH, int kW) {
    const int kWp = (kW + 3) & ~3;
    return ((size_t)(kH + 3) * (64 + kWp - 1 + 3) + (size_t)kH * kWp * 16) * sizeof(float);
}

int dfe_conv_mfma_launch(dfe_ctx *ctx, const float *in, const float *weight, const float *bias, int nIn, int nOut, int H, int W, int kH, int kW,
                         int tanh_after, float *out, bool *handled) {
    *handled = false;
    const size_t lds = conv_mfma_lds_bytes(kH, kW);
    if (lds > 96 * 1024) return DFE_OK;
    const int Ho = H - kH + 1, Wo = W - kW + 1;
    dim3 grid(dfe_cdiv(Wo, 64), dfe_cdiv(Ho, 4), dfe_cdiv(nOut, 16));
    if (tanh_after) {
        DFE_HIP(ctx, hipFuncSetAttribute((const void *)conv_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(conv_mfma_kernel<true>, grid, dim3(256), lds, ctx->stream, in, weight, bias, nIn, nOut, H, W, kH, kW, out);
    } else {
        DFE_HIP(ctx, hipFuncSetAttribute((const void *)conv_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(conv_mfma_kernel<false>, grid, dim3(256), lds, ctx->stream, in, weight, bias, nIn, nOut, H, W, kH, kW, out);
    }
    DFE_LAUNCH_CHECK(ctx);
    ctx->last_kernel = "conv_mfma_kernel";
    *handled = true;
    return DFE_OK;
}

extern "C" {

int dfe_spatial_convolution_mfma_f32(dfe_ctx *ctx, const float *in, const float *weight, const float *bias, int nIn, int nOut, int H, int W,
                                     int kH, int kW, int tanh_after, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, in && weight && out, DFE_E_ARG, "dfe_spatial_convolution_mfma_f32: NULL tensor");
    DFE_REQUIRE(ctx, nIn > 0 && nOut > 0 && kH > 0 && kW > 0 && H >= kH && W >= kW, DFE_E_SHAPE,
                "dfe_spatial_convolution_mfma_f32: %d->%d planes, %dx%d kernel on %dx%d", nIn, nOut, kH, kW, H, W);
    bool handled = false;
    int rc = dfe_conv_mfma_launch(ctx, in, weight, bias, nIn, nOut, H, W, kH, kW, tanh_after, out, &handled);
    if (rc) return rc;
    DFE_REQUIRE(ctx, handled, DFE_E_UNSUPPORTED, "dfe_spatial_convolution_mfma_f32: a %dx%d kernel needs %zu bytes of LDS per block", kH, kW,
                conv_mfma_lds_bytes(kH, kW));
    return DFE_OK;
}

}  // extern "C"
