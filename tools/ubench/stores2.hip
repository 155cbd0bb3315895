// Replays the tiled cost-volume kernel's store ORDER without its compute: a block owns NT=4 tiles of 8
// pixels x TY rows; wave w runs tasks (tile, chunk) t = w, w+8, ...; per task, per row, 8 stores of 64
// floats at pixel stride.  Variants: order A (as the kernel), order B (chunk-major per pixel: a wave writes
// all 18 chunks of a pixel back to back).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void st(const void *base, unsigned off, float v) {
    asm volatile("global_store_dword %0, %1, %2" ::"v"(off), "v"(v), "s"(base) : "memory");
}
template <int ORDER>
__global__ __launch_bounds__(1024) void k(float *out, int Ho, int Wo, int D, int TY, int spin) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nch = (D + 63) / 64;
    const int x0 = min((int)blockIdx.x * 32, Wo - 32), y0 = min((int)blockIdx.y * TY, Ho - TY);
    float v = lane;
    if (ORDER == 0) {
        for (int t = wave; t < 4 * nch; t += 8) {
            int tile = t / nch, chunk = t - tile * nch;
            int d = chunk * 64 + lane;
            if (d < D) {
                for (int r = 0; r < TY; ++r) {
                    for (int s = 0; s < spin; ++s) v = __builtin_fmaf(v, 1.0001f, 0.5f);   // stand-in for the row's VALU work
                    const char *orow = (const char *)(out + ((long long)(y0 + r) * Wo + x0 + tile * 8) * D);
#pragma unroll
                    for (int x = 0; x < 8; ++x) st(orow + (long long)x * D * 4, d * 4u, v);
                }
            }
        }
    } else if (ORDER == 2 || ORDER == 3) {
        // order C: the block walks tiles one at a time; its waves hold ADJACENT chunks (pass*NW + wave) of the
        // same tile and sweep the same rows together (ORDER 3: with a barrier per row)
        const int NW = blockDim.x >> 6;
        for (int tile = 0; tile < 4; ++tile)
            for (int pass = 0; pass * NW < nch; ++pass) {
                int chunk = pass * NW + wave;
                int d = chunk * 64 + lane;
                for (int r = 0; r < TY; ++r) {
                    for (int s = 0; s < spin; ++s) v = __builtin_fmaf(v, 1.0001f, 0.5f);
                    if (ORDER == 3) __syncthreads();
                    if (chunk < nch && d < D) {
                        const char *orow = (const char *)(out + ((long long)(y0 + r) * Wo + x0 + tile * 8) * D);
#pragma unroll
                        for (int x = 0; x < 8; ++x) st(orow + (long long)x * D * 4, d * 4u, v);
                    }
                }
            }
    } else {
        // each wave owns 4 pixels of the 32-wide group per row and writes all chunks of each back to back
        for (int r = 0; r < TY; ++r) {
            for (int px = wave * 4; px < wave * 4 + 4; ++px) {
                const char *op = (const char *)(out + ((long long)(y0 + r) * Wo + x0 + px) * D);
                for (int c = 0; c < nch; ++c) {
                    for (int s = 0; s < spin / 18 * 8 / 4; ++s) v = __builtin_fmaf(v, 1.0001f, 0.5f);
                    int d = c * 64 + lane;
                    if (d < D) st(op, d * 4u, v);
                }
            }
        }
    }
}
template <int ORDER> void run(float *d, int spin, int nw = 8) {
    int Ho = 442, Wo = 602, D = 1089, TY = 18;
    dim3 grid((Wo + 31) / 32, (Ho + TY - 1) / TY);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<ORDER>, grid, dim3(nw * 64), 0, 0, d, Ho, Wo, D, TY, spin);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("order %d nw=%d spin=%4d: %.3f ms  %.2f TB/s\n", ORDER, nw, spin, ms, 442.0 * 602 * 1089 * 4 / ms / 1e9);
}
int main() {
    float *d; (void)hipMalloc(&d, 442ll * 602 * 1089 * 4 + 4096);
    run<0>(d, 0); run<1>(d, 0);
    for (int nw : {6, 9, 8, 16}) { run<2>(d, 0, nw); run<3>(d, 0, nw); }
    return 0;
}
