// VALU issue-rate microbenchmark for gfx950: scalar v_fma_f32 vs packed v_pk_fma_f32, by waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ void k(float *out, int iters, float a, float b) {
    if constexpr (PK == 0) {
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            }
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    } else {
        f2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
        f2 av = {a, a}, bv = {b, b};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = __builtin_elementwise_fma(x0, av, bv); x1 = __builtin_elementwise_fma(x1, av, bv);
                x2 = __builtin_elementwise_fma(x2, av, bv); x3 = __builtin_elementwise_fma(x3, av, bv);
                x4 = __builtin_elementwise_fma(x4, av, bv); x5 = __builtin_elementwise_fma(x5, av, bv);
                x6 = __builtin_elementwise_fma(x6, av, bv); x7 = __builtin_elementwise_fma(x7, av, bv);
            }
        }
        f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
        out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
    }
}
template <int PK> void run(float *d, int wps) {
    int iters = 4096;
    int blocks = 256 * wps;  // 256-thread blocks = 1 wave per SIMD each
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<PK>, dim3(blocks), dim3(256), 0, 0, d, 64, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<PK>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters * 64;     // wave-instructions
    double per_cu_per_us = winstr / 256 / (ms * 1e3);
    printf("pk=%d waves/SIMD=%d  %.3f ms  %.0f wave-instr/us/CU (= %.2f per cycle @2.4GHz)  %.1f TFLOP/s\n", PK, wps, ms,
           per_cu_per_us, per_cu_per_us / 2400.0, winstr * 64 * 2 * (PK ? 2 : 1) / (ms * 1e-3) / 1e12);
}
int main() {
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4 * sizeof(float));
    for (int w : {1, 2, 3, 4, 6, 8}) run<0>(d, w);
    for (int w : {1, 2, 3, 4, 6, 8}) run<1>(d, w);
    return 0;
}
