// Which factor hurts the row-image copy-out: the barrier burst or the interleaving of 16 waves' 1-KB pieces?
// Every wave writes its two line-aligned 1-KB pieces (dwordx4) of each row image (8 runs x 4 KB); variants:
// barrier per row or free-running; optional VALU filler between rows to emulate the compute phase.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(float *out, int Ho, int Wo, int D, int TY, int sync, int spin, int nost) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int x0 = min((int)blockIdx.x * 8, Wo - 8), y0 = min((int)blockIdx.y * TY, Ho - TY);
    float4 v = make_float4(lane, 1, 2, 3);
    float f0 = lane, f1 = lane + 1, f2 = lane + 2, f3 = lane + 3;
    for (int r = 0; r < TY; ++r) {
        for (int s = 0; s < spin; ++s) {   // 4 independent FMA chains ~ the row's VALU work
            f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, 0.5f);
            f2 = __builtin_fmaf(f2, 1.0001f, 0.5f); f3 = __builtin_fmaf(f3, 1.0001f, 0.5f);
        }
        if (sync) __syncthreads();
        const long long G0 = ((long long)(y0 + r) * Wo + x0) * D;
        const int a0 = (int)(G0 & 31);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int slot = tid + 1024 * k2, x = slot >> 8, j = slot & 255;
            const int ax = (a0 + x * D) & 31, head = (32 - ax) & 31;
            const int nb4 = ((1024 - head) >> 5) << 3;
            if (j < nb4 && (!nost || f0 == -3.f)) reinterpret_cast<float4 *>(out + G0 + (long long)x * D + head)[j] = v;
        }
    }
    if (f0 + f1 + f2 + f3 == -1.f) out[0] = f0;
}
void run(float *d, int sync, int spin, int nost = 0) {
    int Ho = 442, Wo = 602, D = 1089, TY = 24;
    dim3 grid((Wo + 7) / 8, (Ho + TY - 1) / TY);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, grid, dim3(1024), 100*1024, 0, d, Ho, Wo, D, TY, sync, spin, nost);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("all 16 waves copy 2 aligned KB each per row, sync=%d spin=%3d nostore=%d: %.3f ms  %.2f TB/s\n", sync, spin, nost, ms, 442.0 * 602 * 1024 * 4 / ms / 1e9);
}
int main() {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100*1024);
    float *d; (void)hipMalloc(&d, 442ll * 602 * 1089 * 4 + 4096);
    for (int spin : {0, 20, 30, 40, 60}) { run(d, 1, spin); run(d, 1, spin, 1); }
    return 0;
}
