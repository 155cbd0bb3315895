// Store pattern of the LDS-exchange kernel: per row a block of NW waves writes, for each of TX pixels, one
// contiguous run of NW*64 floats (pass p of the pixel's D floats); wave w writes pieces w*TX..w*TX+TX-1 of the
// (pixel, chunk) order back to back.  RUNS=1: all passes of a row are written in the same row step (full pixel).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void st(const void *base, unsigned off, float v) {
    asm volatile("global_store_dword %0, %1, %2" ::"v"(off), "v"(v), "s"(base) : "memory");
}
template <int NW, int FULL>
__global__ __launch_bounds__(NW * 64) void k(float *out, int Ho, int Wo, int D, int TY, int sync) {
    constexpr int TX = 8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nch = (D + 63) / 64, npass = (nch + NW - 1) / NW;
    const int x0 = min((int)blockIdx.x * TX, Wo - TX), y0 = min((int)blockIdx.y * TY, Ho - TY);
    float v = lane;
    if (FULL) {
        for (int r = 0; r < TY; ++r) {
            if (sync) __syncthreads();
            for (int pass = 0; pass < npass; ++pass) {
                const char *orow = (const char *)(out + ((long long)(y0 + r) * Wo + x0) * D) + (long long)pass * NW * 256;
#pragma unroll
                for (int kk = 0; kk < TX; ++kk) {
                    int piece = wave * TX + kk, x = piece / NW, c = piece - x * NW, ch = pass * NW + c;
                    if (ch * 64 + lane < D) st(orow + ((long long)x * D + c * 64) * 4, lane * 4u, v);
                }
            }
        }
    } else {
        for (int pass = 0; pass < npass; ++pass)
            for (int r = 0; r < TY; ++r) {
                if (sync) __syncthreads();
                const char *orow = (const char *)(out + ((long long)(y0 + r) * Wo + x0) * D) + (long long)pass * NW * 256;
#pragma unroll
                for (int kk = 0; kk < TX; ++kk) {
                    int piece = wave * TX + kk, x = piece / NW, c = piece - x * NW, ch = pass * NW + c;
                    if (ch * 64 + lane < D) st(orow + ((long long)x * D + c * 64) * 4, lane * 4u, v);
                }
            }
    }
}
template <int NW, int FULL> void run(float *d, int sync) {
    int Ho = 442, Wo = 602, D = 1089, TY = 18;
    dim3 grid((Wo + 7) / 8, (Ho + TY - 1) / TY);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NW, FULL>), grid, dim3(NW * 64), 0, 0, d, Ho, Wo, D, TY, sync);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("NW=%d %s sync=%d: %.3f ms  %.2f TB/s\n", NW, FULL ? "full-pixel rows" : "per-pass runs  ", sync, ms, 442.0 * 602 * 1089 * 4 / ms / 1e9);
}
int main() {
    float *d; (void)hipMalloc(&d, 442ll * 602 * 1089 * 4 + 4096);
    run<6, 0>(d, 0); run<6, 0>(d, 1); run<9, 0>(d, 1); run<6, 1>(d, 0); run<6, 1>(d, 1); run<9, 1>(d, 1); run<8, 1>(d, 1);
    return 0;
}
