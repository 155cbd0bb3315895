// matching_backward.hip -- gradients of nn.SpatialMatching / nn.SpatialRadialMatching w.r.t. both feature maps (next-row N2:
// the training drivers radial/train_radial_opticalflow.lua:228-252 and opticalflow.lua:296-338 call model:backward through
// the matcher).  Gather form: one thread per input element sums its own window of terms in (dy, dx) order -- no atomics,
// bit-identical to the CPU restatement.  Patch-sized and latency-bound in training; not tuned further.
#include "dfe_internal.h"

namespace {

struct MatchGeom {
    int K, H1, W1, maxh, maxw, H2, W2;
};

// g1[k][y][x] = sum_{dy,dx} 2 (in1[k][y][x] - in2[k][y+dy][x+dx]) go[y][x][dy][dx]
__global__ void matching_grad1_kernel(MatchGeom g, const float *__restrict__ in1, const float *__restrict__ in2,
                                      const float *__restrict__ go, float *__restrict__ g1) {
#pragma clang fp contract(off)
    const long long n = (long long)g.K * g.H1 * g.W1;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(e % g.W1);
        const long long t = e / g.W1;
        const int y = (int)(t % g.H1), k = (int)(t / g.H1);
        const float a = in1[e];
        const float *b = in2 + ((long long)k * g.H2 + y) * g.W2 + x;
        const float *w = go + ((long long)y * g.W1 + x) * g.maxh * g.maxw;
        float s = 0.f;
        for (int dy = 0; dy < g.maxh; ++dy)
            for (int dx = 0; dx < g.maxw; ++dx) s = s + 2.0f * (a - b[(long long)dy * g.W2 + dx]) * w[dy * g.maxw + dx];
        g1[e] = s;
    }
}

// g2[k][v][u] = sum_{dy,dx} -2 (in1[k][v-dy][u-dx] - in2[k][v][u]) go[v-dy][u-dx][dy][dx], (v-dy, u-dx) inside in1
__global__ void matching_grad2_kernel(MatchGeom g, const float *__restrict__ in1, const float *__restrict__ in2,
                                      const float *__restrict__ go, float *__restrict__ g2) {
#pragma clang fp contract(off)
    const long long n = (long long)g.K * g.H2 * g.W2;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int u = (int)(e % g.W2);
        const long long t = e / g.W2;
        const int v = (int)(t % g.H2), k = (int)(t / g.H2);
        const float b = in2[e];
        float s = 0.f;
        for (int dy = 0; dy < g.maxh; ++dy) {
            const int y = v - dy;
            if (y < 0 || y >= g.H1) continue;
            for (int dx = 0; dx < g.maxw; ++dx) {
                const int x = u - dx;
                if (x < 0 || x >= g.W1) continue;
                s = s + -2.0f * (in1[((long long)k * g.H1 + y) * g.W1 + x] - b) *
                            go[(((long long)y * g.W1 + x) * g.maxh + dy) * g.maxw + dx];
            }
        }
        g2[e] = s;
    }
}

int grid_for_n(long long n) {
    long long b = (n + 255) / 256;
    if (b > 256 * 32) b = 256 * 32;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" {

int dfe_spatial_matching_backward_f32(dfe_ctx *ctx, const float *in1, const float *in2, const float *gradOut, int K, int H1, int W1,
                                      int maxh, int maxw, float *gradIn1, float *gradIn2) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, in1 && in2 && gradOut && (gradIn1 || gradIn2), DFE_E_ARG, "dfe_spatial_matching_backward_f32: NULL tensor");
    DFE_REQUIRE(ctx, K > 0 && H1 > 0 && W1 > 0 && maxh > 0 && maxw > 0, DFE_E_SHAPE,
                "dfe_spatial_matching_backward_f32: K=%d H1=%d W1=%d window %dx%d must be positive", K, H1, W1, maxh, maxw);
    MatchGeom g{K, H1, W1, maxh, maxw, H1 + maxh - 1, W1 + maxw - 1};
    if (gradIn1)
        hipLaunchKernelGGL(matching_grad1_kernel, dim3(grid_for_n((long long)K * H1 * W1)), dim3(256), 0, ctx->stream, g, in1, in2, gradOut,
                           gradIn1);
    if (gradIn2)
        hipLaunchKernelGGL(matching_grad2_kernel, dim3(grid_for_n((long long)K * g.H2 * g.W2)), dim3(256), 0, ctx->stream, g, in1, in2,
                           gradOut, gradIn2);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_radial_matching_backward_f32(dfe_ctx *ctx, const float *in1, const float *in2, const float *gradOut, int K, int H1, int W, int hWin,
                                     float *gradIn1, float *gradIn2) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, hWin > 0, DFE_E_SHAPE, "dfe_radial_matching_backward_f32: hWin=%d must be positive", hWin);
    return dfe_spatial_matching_backward_f32(ctx, in1, in2, gradOut, K, H1, W, hWin, 1, gradIn1, gradIn2);
}

}  // extern "C"
