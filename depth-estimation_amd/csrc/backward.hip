// backward.hip -- next-row N2: the gradients that let `network:backward(input, df_do)` reach the filter weights
// (radial/train_radial_opticalflow.lua:228-252, opticalflow.lua:296-338) through the drop-in modules:
//   nn.SpatialConvolution / nn.SpatialConvolutionMap   updateGradInput + accGradParameters   (un-vendored nn; pinned here, as the
//                                                       reference pins its own module, by the Jacobian of the forward:
//                                                       tests/test_cascad.lua:21-25 method)
//   nn.Tanh                                            gradIn = gradOut * (1 - out^2)
//   nn.Log2                                            forward: clamp the input IN PLACE to >= eps, then log (Log.lua:13-22);
//                                                      backward: gradOut / input (Log.lua:24-28)
//   nn.SoftMax over the window (getModel's FunctionWrapper, opticalflow_model.lua:96-109) backward
//   nn.LogSoftMax (radial trainer, radial_opticalflow_network.lua:50) forward + backward
// Gradients w.r.t. inputs are gathers (one thread per input element, fixed term order o, u, v -> bit-identical to the
// oracle's loop); parameter gradients are block reductions over the output pixels (tolerance 1e-5 relative against the
// oracle's sequential sum).
#include "dfe_internal.h"

namespace {

int grid_n(long long n) {
    long long b = (n + 255) / 256;
    if (b > 256 * 32) b = 256 * 32;
    if (b < 1) b = 1;
    return (int)b;
}

// gradIn[i][y][x] = sum_o sum_u sum_v w[o][i][u][v] * gO[o][y-u][x-v]
__global__ void conv_grad_input_kernel(const float *__restrict__ go, const float *__restrict__ w, int nIn, int nOut, int H, int W, int kH, int kW,
                                       float *__restrict__ gi) {
#pragma clang fp contract(off)
    const int Ho = H - kH + 1, Wo = W - kW + 1;
    const long long n = (long long)nIn * H * W;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(e % W);
        const long long t = e / W;
        const int y = (int)(t % H), i = (int)(t / H);
        float s = 0.f;
        for (int o = 0; o < nOut; ++o)
            for (int u = 0; u < kH; ++u) {
                const int yy = y - u;
                if (yy < 0 || yy >= Ho) continue;
                for (int v = 0; v < kW; ++v) {
                    const int xx = x - v;
                    if (xx < 0 || xx >= Wo) continue;
                    s = s + w[(((long long)o * nIn + i) * kH + u) * kW + v] * go[((long long)o * Ho + yy) * Wo + xx];
                }
            }
        gi[e] = s;
    }
}

// the same over a connection table: plane i receives from every connection (i -> o)
__global__ void conv_map_grad_input_kernel(const float *__restrict__ go, const float *__restrict__ w, const int *__restrict__ conn, int nConn,
                                           int nIn, int H, int W, int kH, int kW, float *__restrict__ gi) {
#pragma clang fp contract(off)
    const int Ho = H - kH + 1, Wo = W - kW + 1;
    const long long n = (long long)nIn * H * W;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(e % W);
        const long long t = e / W;
        const int y = (int)(t % H), i = (int)(t / H);
        float s = 0.f;
        for (int c = 0; c < nConn; ++c) {
            if (conn[2 * c] - 1 != i) continue;
            const int o = conn[2 * c + 1] - 1;
            for (int u = 0; u < kH; ++u) {
                const int yy = y - u;
                if (yy < 0 || yy >= Ho) continue;
                for (int v = 0; v < kW; ++v) {
                    const int xx = x - v;
                    if (xx < 0 || xx >= Wo) continue;
                    s = s + w[((long long)c * kH + u) * kW + v] * go[((long long)o * Ho + yy) * Wo + xx];
                }
            }
        }
        gi[e] = s;
    }
}

__device__ __forceinline__ float block_sum_256(float v, float *sm) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sm[w] = v;
    __syncthreads();
    float r = sm[0] + sm[1] + sm[2] + sm[3];
    __syncthreads();
    return r;
}

// one block per weight: gW[o][i][u][v] += scale * sum_{y,x} gO[o][y][x] * in[i][y+u][x+v]   (conn == NULL: dense layout)
__global__ __launch_bounds__(256) void conv_acc_grad_weight_kernel(const float *__restrict__ in, const float *__restrict__ go,
                                                                  const int *__restrict__ conn, int nIn, int H, int W, int kH, int kW,
                                                                  float scale, float *__restrict__ gw) {
    __shared__ float sm[4];
    const int Ho = H - kH + 1, Wo = W - kW + 1;
    const int wi = blockIdx.x;                     // flat weight index
    const int v = wi % kW, u = (wi / kW) % kH, q = wi / (kW * kH);
    int o, i;
    if (conn) { i = conn[2 * q] - 1; o = conn[2 * q + 1] - 1; }
    else { i = q % nIn; o = q / nIn; }
    const float *gp = go + (long long)o * Ho * Wo, *ip = in + ((long long)i * H + u) * W + v;
    float s = 0.f;
    for (int p = threadIdx.x; p < Ho * Wo; p += 256) {
        const int y = p / Wo, x = p - y * Wo;
        s += gp[p] * ip[(long long)y * W + x];
    }
    const float t = block_sum_256(s, sm);
    if (threadIdx.x == 0) gw[wi] += scale * t;
}

__global__ __launch_bounds__(256) void conv_acc_grad_bias_kernel(const float *__restrict__ go, int P, float scale, float *__restrict__ gb) {
    __shared__ float sm[4];
    const float *gp = go + (long long)blockIdx.x * P;
    float s = 0.f;
    for (int p = threadIdx.x; p < P; p += 256) s += gp[p];
    const float t = block_sum_256(s, sm);
    if (threadIdx.x == 0) gb[blockIdx.x] += scale * t;
}

__global__ void tanh_backward_kernel(const float *__restrict__ out, const float *__restrict__ go, long long n, float *__restrict__ gi) {
#pragma clang fp contract(off)
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const float o = out[e];
        gi[e] = go[e] * (1.f - o * o);
    }
}

__global__ void log_clamp_kernel(float *__restrict__ in, long long n, float eps, int clamp, float *__restrict__ out) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        float x = in[e];
        if (clamp && x < eps) { x = eps; in[e] = x; }   // Log.lua:15-18: the clamped value is written back into the input
        out[e] = logf(x);
    }
}

__global__ void div_kernel(const float *__restrict__ a, const float *__restrict__ b, long long n, float *__restrict__ out) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) out[e] = a[e] / b[e];
}

// rows of N: one wave per row, lanes stride the row (N small: 15 .. 1089)
__device__ __forceinline__ float wave_max(float v) {
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(256) void log_softmax_kernel(const float *__restrict__ in, long long P, int N, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); p < P; p += (long long)gridDim.x * 4) {
        const float *r = in + p * N;
        float m = -INFINITY;
        for (int j = lane; j < N; j += 64) m = fmaxf(m, r[j]);
        m = wave_max(m);
        float s = 0.f;
        for (int j = lane; j < N; j += 64) s += expf(r[j] - m);
        s = wave_sum(s);
        const float l = m + logf(s);
        for (int j = lane; j < N; j += 64) out[p * N + j] = r[j] - l;
    }
}

// gradIn = gradOut - exp(out) * sum(gradOut)
__global__ __launch_bounds__(256) void log_softmax_backward_kernel(const float *__restrict__ out, const float *__restrict__ go, long long P, int N,
                                                                  float *__restrict__ gi) {
    const int lane = threadIdx.x & 63;
    for (long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); p < P; p += (long long)gridDim.x * 4) {
        float s = 0.f;
        for (int j = lane; j < N; j += 64) s += go[p * N + j];
        s = wave_sum(s);
        for (int j = lane; j < N; j += 64) gi[p * N + j] = go[p * N + j] - expf(out[p * N + j]) * s;
    }
}

// softmax backward: gradIn = out * (gradOut - sum(gradOut * out))
__global__ __launch_bounds__(256) void softmax_backward_kernel(const float *__restrict__ out, const float *__restrict__ go, long long P, int N,
                                                              float *__restrict__ gi) {
    const int lane = threadIdx.x & 63;
    for (long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); p < P; p += (long long)gridDim.x * 4) {
        float s = 0.f;
        for (int j = lane; j < N; j += 64) s += go[p * N + j] * out[p * N + j];
        s = wave_sum(s);
        for (int j = lane; j < N; j += 64) gi[p * N + j] = out[p * N + j] * (go[p * N + j] - s);
    }
}

}  // namespace

extern "C" {

int dfe_spatial_convolution_grad_input_f32(dfe_ctx *ctx, const float *gradOut, const float *weight, int nIn, int nOut, int H, int W, int kH,
                                           int kW, float *gradIn) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, gradOut && weight && gradIn, DFE_E_ARG, "dfe_spatial_convolution_grad_input_f32: NULL tensor");
    DFE_REQUIRE(ctx, nIn > 0 && nOut > 0 && kH > 0 && kW > 0 && H >= kH && W >= kW, DFE_E_SHAPE,
                "dfe_spatial_convolution_grad_input_f32: %d->%d planes, %dx%d kernel on %dx%d", nIn, nOut, kH, kW, H, W);
    hipLaunchKernelGGL(conv_grad_input_kernel, dim3(grid_n((long long)nIn * H * W)), dim3(256), 0, ctx->stream, gradOut, weight, nIn, nOut, H, W,
                       kH, kW, gradIn);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_spatial_convolution_acc_grad_f32(dfe_ctx *ctx, const float *in, const float *gradOut, int nIn, int nOut, int H, int W, int kH, int kW,
                                         float scale, float *gradWeight, float *gradBias) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, in && gradOut && gradWeight, DFE_E_ARG, "dfe_spatial_convolution_acc_grad_f32: NULL tensor");
    DFE_REQUIRE(ctx, nIn > 0 && nOut > 0 && kH > 0 && kW > 0 && H >= kH && W >= kW, DFE_E_SHAPE,
                "dfe_spatial_convolution_acc_grad_f32: %d->%d planes, %dx%d kernel on %dx%d", nIn, nOut, kH, kW, H, W);
    hipLaunchKernelGGL(conv_acc_grad_weight_kernel, dim3(nOut * nIn * kH * kW), dim3(256), 0, ctx->stream, in, gradOut, (const int *)nullptr, nIn,
                       H, W, kH, kW, scale, gradWeight);
    if (gradBias)
        hipLaunchKernelGGL(conv_acc_grad_bias_kernel, dim3(nOut), dim3(256), 0, ctx->stream, gradOut, (H - kH + 1) * (W - kW + 1), scale, gradBias);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_spatial_convolution_map_grad_input_f32(dfe_ctx *ctx, const float *gradOut, const float *weight, const int32_t *conn, int nConn, int nIn,
                                               int nOut, int H, int W, int kH, int kW, float *gradIn) {
    DFE_ENTER(ctx);
    (void)nOut;
    DFE_REQUIRE(ctx, gradOut && weight && conn && gradIn, DFE_E_ARG, "dfe_spatial_convolution_map_grad_input_f32: NULL tensor");
    DFE_REQUIRE(ctx, nIn > 0 && nConn > 0 && kH > 0 && kW > 0 && H >= kH && W >= kW, DFE_E_SHAPE,
                "dfe_spatial_convolution_map_grad_input_f32: %d connections, %dx%d kernel on %dx%d", nConn, kH, kW, H, W);
    hipLaunchKernelGGL(conv_map_grad_input_kernel, dim3(grid_n((long long)nIn * H * W)), dim3(256), 0, ctx->stream, gradOut, weight, conn, nConn,
                       nIn, H, W, kH, kW, gradIn);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_spatial_convolution_map_acc_grad_f32(dfe_ctx *ctx, const float *in, const float *gradOut, const int32_t *conn, int nConn, int nIn,
                                             int nOut, int H, int W, int kH, int kW, float scale, float *gradWeight, float *gradBias) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, in && gradOut && conn && gradWeight, DFE_E_ARG, "dfe_spatial_convolution_map_acc_grad_f32: NULL tensor");
    DFE_REQUIRE(ctx, nIn > 0 && nOut > 0 && nConn > 0 && kH > 0 && kW > 0 && H >= kH && W >= kW, DFE_E_SHAPE,
                "dfe_spatial_convolution_map_acc_grad_f32: %d connections %d->%d planes, %dx%d kernel on %dx%d", nConn, nIn, nOut, kH, kW, H, W);
    hipLaunchKernelGGL(conv_acc_grad_weight_kernel, dim3(nConn * kH * kW), dim3(256), 0, ctx->stream, in, gradOut, conn, nIn, H, W, kH, kW, scale,
                       gradWeight);
    if (gradBias)
        hipLaunchKernelGGL(conv_acc_grad_bias_kernel, dim3(nOut), dim3(256), 0, ctx->stream, gradOut, (H - kH + 1) * (W - kW + 1), scale, gradBias);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_tanh_backward_f32(dfe_ctx *ctx, const float *out, const float *gradOut, int64_t n, float *gradIn) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, n >= 0, DFE_E_SHAPE, "dfe_tanh_backward_f32: n=%lld", (long long)n);
    if (n == 0) return DFE_OK;
    DFE_REQUIRE(ctx, out && gradOut && gradIn, DFE_E_ARG, "dfe_tanh_backward_f32: NULL tensor");
    hipLaunchKernelGGL(tanh_backward_kernel, dim3(grid_n(n)), dim3(256), 0, ctx->stream, out, gradOut, (long long)n, gradIn);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_log2_forward_f32(dfe_ctx *ctx, float *input, int64_t n, float null_epsilon, int clamp, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, n >= 0, DFE_E_SHAPE, "dfe_log2_forward_f32: n=%lld", (long long)n);
    if (n == 0) return DFE_OK;
    DFE_REQUIRE(ctx, input && out, DFE_E_ARG, "dfe_log2_forward_f32: NULL tensor");
    hipLaunchKernelGGL(log_clamp_kernel, dim3(grid_n(n)), dim3(256), 0, ctx->stream, input, (long long)n, null_epsilon, clamp, out);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_log2_backward_f32(dfe_ctx *ctx, const float *input, const float *gradOut, int64_t n, float *gradIn) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, n >= 0, DFE_E_SHAPE, "dfe_log2_backward_f32: n=%lld", (long long)n);
    if (n == 0) return DFE_OK;
    DFE_REQUIRE(ctx, input && gradOut && gradIn, DFE_E_ARG, "dfe_log2_backward_f32: NULL tensor");
    hipLaunchKernelGGL(div_kernel, dim3(grid_n(n)), dim3(256), 0, ctx->stream, gradOut, input, (long long)n, gradIn);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_log_softmax_f32(dfe_ctx *ctx, const float *in, int64_t P, int N, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, P >= 0 && N > 0, DFE_E_SHAPE, "dfe_log_softmax_f32: P=%lld N=%d", (long long)P, N);
    if (P == 0) return DFE_OK;
    DFE_REQUIRE(ctx, in && out, DFE_E_ARG, "dfe_log_softmax_f32: NULL tensor");
    hipLaunchKernelGGL(log_softmax_kernel, dim3(grid_n(P * 64)), dim3(256), 0, ctx->stream, in, (long long)P, N, out);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_log_softmax_backward_f32(dfe_ctx *ctx, const float *out, const float *gradOut, int64_t P, int N, float *gradIn) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, P >= 0 && N > 0, DFE_E_SHAPE, "dfe_log_softmax_backward_f32: P=%lld N=%d", (long long)P, N);
    if (P == 0) return DFE_OK;
    DFE_REQUIRE(ctx, out && gradOut && gradIn, DFE_E_ARG, "dfe_log_softmax_backward_f32: NULL tensor");
    hipLaunchKernelGGL(log_softmax_backward_kernel, dim3(grid_n(P * 64)), dim3(256), 0, ctx->stream, out, gradOut, (long long)P, N, gradIn);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_softmax_backward_f32(dfe_ctx *ctx, const float *out, const float *gradOut, int64_t P, int N, float *gradIn) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, P >= 0 && N > 0, DFE_E_SHAPE, "dfe_softmax_backward_f32: P=%lld N=%d", (long long)P, N);
    if (P == 0) return DFE_OK;
    DFE_REQUIRE(ctx, out && gradOut && gradIn, DFE_E_ARG, "dfe_softmax_backward_f32: NULL tensor");
    hipLaunchKernelGGL(softmax_backward_kernel, dim3(grid_n(P * 64)), dim3(256), 0, ctx->stream, out, gradOut, (long long)P, N, gradIn);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

}  // extern "C"
