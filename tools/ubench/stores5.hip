// Candidate store pattern: a block of 16 waves, per row, writes for each of 8 pixels one contiguous run of RUNF
// floats (of the pixel's 1089) with 16-B-per-lane stores, waves 4x..4x+3 covering pixel x (k=0) / x+4 (k=1).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int RUNF>
__global__ __launch_bounds__(1024) void k(float *out, int Ho, int Wo, int D, int TY, int sync) {
    const int tid = threadIdx.x;
    const int x0 = min((int)blockIdx.x * 8, Wo - 8), y0 = min((int)blockIdx.y * TY, Ho - TY);
    float4 v = make_float4(tid, 1, 2, 3);
    constexpr int F4 = RUNF / 4;          // float4 per run
    for (int r = 0; r < TY; ++r) {
        if (sync) __syncthreads();
        float *orow = out + ((long long)(y0 + r) * Wo + x0) * D;
        for (int i = tid; i < 8 * F4; i += 1024) {
            int x = i / F4, j = i - x * F4;
            __builtin_memcpy(orow + (long long)x * D + j * 4, &v, 16);   // 4-B aligned 16-B store
        }
    }
}
template <int RUNF> void run(float *d, int sync) {
    int Ho = 442, Wo = 602, D = 1089, TY = 18;
    dim3 grid((Wo + 7) / 8, (Ho + TY - 1) / TY);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<RUNF>, grid, dim3(1024), 0, 0, d, Ho, Wo, D, TY, sync);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("16 waves, runs of %d floats, x4 stores, sync=%d: %.3f ms  %.2f TB/s (of the bytes it writes)\n", RUNF, sync, ms,
           442.0 * 602 * RUNF * 4 / ms / 1e9);
}
int main() {
    float *d; (void)hipMalloc(&d, 442ll * 602 * 1089 * 4 + 4096);
    run<1024>(d, 1); run<1024>(d, 0); run<1088>(d, 1); run<512>(d, 1);
    return 0;
}
