// Tuning only: what the f32 matrix pipe delivers.  v_mfma_f32_16x16x4_f32 on 8 independent accumulators per wave, 16 waves per CU, with
// (a) register operands only, (b) one ds_read_b32 per MFMA (the matcher's / the convolution's operand delivery), (c) two accumulators only
// (the convolution's chain distance).  Reports TFLOP/s, s_memtime ticks per ns and ticks per MFMA and SIMD.
// build: hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));
template <int NACC, bool LDS>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *ticks, int iters) {
    __shared__ float sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 1024) sm[i] = (float)i * 1e-6f;
    __syncthreads();
    f4v acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = f4v{0.f, 0.f, 0.f, 0.f};
    float av = threadIdx.x * 1e-3f, bv = 1.0f;
    const float *lp = sm + (threadIdx.x & 63);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            float b = bv;
            if constexpr (LDS) b = lp[((it * NACC + a) & 31) * 64];
            acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[a], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
// random operands (8 register pairs per lane, values of mixed magnitude and sign): does the data decide the clock?
template <int NACC>
__global__ __launch_bounds__(1024) void krand(const float *__restrict__ rnd, float *out, unsigned long long *ticks, int iters) {
    f4v acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = f4v{0.f, 0.f, 0.f, 0.f};
    float av[8], bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { av[j] = rnd[(threadIdx.x * 16 + j) & 65535]; bv[j] = rnd[(threadIdx.x * 16 + 8 + j + blockIdx.x) & 65535]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(av[j]), "+v"(bv[j]));   // (the loads are waited for HERE, not inside the loop)
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 8 / NACC > 0 ? 1 : 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc[j % NACC], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int NACC> void run_rand(const char *name, float scale) {
    const int blocks = 256, iters = 8192;
    float *out, *rnd; unsigned long long *ticks;
    CK(hipMalloc(&out, blocks * 1024 * 4)); CK(hipMalloc(&ticks, blocks * 8)); CK(hipMalloc(&rnd, 65536 * 4));
    float *h = (float *)malloc(65536 * 4);
    unsigned x = 12345;
    for (int i = 0; i < 65536; ++i) { x = x * 1664525u + 1013904223u; h[i] = scale == 0.f ? 0.f : ((int)(x >> 8) - (1 << 23)) * scale / (float)(1 << 23); }
    CK(hipMemcpy(rnd, h, 65536 * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((krand<NACC>), dim3(blocks), dim3(1024), 0, 0, rnd, out, ticks, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double nmfma = (double)blocks * 16 * iters * 8;
        printf("%-44s waves/CU 16: %.3f ms  %.1f TFLOP/s\n", name, ms, nmfma * 2048 / ms * 1e-9);
    }
    CK(hipFree(out)); CK(hipFree(ticks)); CK(hipFree(rnd)); free(h);
}
template <int NACC, bool LDS> void run(const char *name, int waves_per_block) {
    const int blocks = 256, iters = 8192 * 8 / NACC;
    float *out; unsigned long long *ticks;
    CK(hipMalloc(&out, blocks * 1024 * 4)); CK(hipMalloc(&ticks, blocks * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(waves_per_block * 64), 0, 0, out, ticks, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        unsigned long long h[256]; CK(hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost));
        double tk = 0; for (int i = 0; i < blocks; ++i) tk += (double)h[i]; tk /= blocks;
        const double nmfma = (double)blocks * waves_per_block * iters * NACC;
        const double per_simd = nmfma / 1024.0;
        printf("%-44s waves/CU %2d: %.3f ms  %.1f TFLOP/s  ticks/ns %.3f  ticks per MFMA and SIMD %.1f\n", name, waves_per_block, ms, nmfma * 2048 / ms * 1e-9, tk / (ms * 1e6),
               tk / per_simd);
    }
    CK(hipFree(out)); CK(hipFree(ticks));
}
int main() {
    run<8, false>("8 accumulators, register operands", 16);
    run<8, false>("8 accumulators, register operands", 4);
    run<8, true>("8 accumulators, one ds_read_b32 per MFMA", 16);
    run<2, false>("2 accumulators, register operands", 16);
    run<2, true>("2 accumulators, one ds_read_b32 per MFMA", 16);
    run<2, false>("2 accumulators, register operands", 4);
    run_rand<2>("2 accumulators, 8 operand pairs, all zero", 0.f);
    run_rand<2>("2 accumulators, 8 operand pairs, random +-1", 1.f);
    run_rand<8>("8 accumulators, 8 operand pairs, random +-1", 1.f);
    run_rand<2>("2 accumulators, 8 operand pairs, random +-1e-3", 1e-3f);
    run_rand<4>("4 accumulators, 8 operand pairs, random +-1", 1.f);
    run_rand<1>("1 accumulator, 8 operand pairs, random +-1", 1.f);
    return 0;
}
