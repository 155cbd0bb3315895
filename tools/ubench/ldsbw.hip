// LDS read bandwidth per CU by access width, 16 waves per CU, every lane reading consecutive 16 / 8 / 4 bytes (the flat matcher's
// operand reads: a lane's window row = 5 x ds_read_b128 at lane stride 16 B).  hipcc --offload-arch=gfx950 -O3 -o ldsbw ldsbw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f4 lds_f4;
typedef __attribute__((address_space(3))) f2 lds_f2;
typedef __attribute__((address_space(3))) float lds_f;
extern __shared__ __attribute__((aligned(128))) float smem[];
template <int W, int STRIDE16> __global__ __launch_bounds__(1024) void k(float *out, int iters) {
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) smem[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    const unsigned base = (unsigned)(size_t)(lds_f *)smem + wave * 1280 * 4 + lane * (STRIDE16 ? 16 : W * 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (W == 4) { f4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(u * 16)); asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); acc += v[0]; }
            if (W == 2) { f2 v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(u * 16)); asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); acc += v[0]; }
            if (W == 1) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(u * 16)); asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); acc += v; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int W, int S16> void run(const char *name, int threads) {
    float *d; (void)hipMalloc(&d, 256 * 1024 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void *)k<W, S16>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    float ms = 0;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<W, S16>), dim3(256), dim3(threads), 128 * 1024, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double reads = (double)iters * 8 * (threads / 64);
    const double bytes = reads * 64 * W * 4;
    printf("%-34s %2d waves/CU: %.3f ms, %.1f ns per wave-read and CU = %.1f cycles at 2.1 GHz, %.0f B per cycle and CU\n", name, threads / 64, ms, ms * 1e6 / reads,
           ms * 1e6 / reads * 2.1, bytes / (ms * 1e6 * 2.1));
    (void)hipFree(d);
}
int main() {
    run<4, 0>("ds_read_b128, lanes 16 B apart", 1024);
    run<4, 0>("ds_read_b128, lanes 16 B apart", 512);
    run<4, 0>("ds_read_b128, lanes 16 B apart", 256);
    run<2, 0>("ds_read_b64, lanes 8 B apart", 1024);
    run<2, 1>("ds_read_b64, lanes 16 B apart", 1024);
    run<1, 0>("ds_read_b32, lanes 4 B apart", 1024);
    run<1, 1>("ds_read_b32, lanes 16 B apart", 1024);
    return 0;
}
