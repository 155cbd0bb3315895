// The flat matcher's plane arithmetic by itself (csrc/feat_matching_flat.hip: 4 pixels x 16 cells per lane, difference / square / add with
// separately rounded operations, batches of 8), on register data: what the vector ALU alone needs per plane with 16 waves per CU.
// hipcc --offload-arch=gfx950 -O3 -o fmix fmix.hip && ./fmix
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MW, int BATCH> __global__ __launch_bounds__(1024) void k(float *out, int planes, float seed) {
#pragma clang fp contract(off)
    float acc[4][MW], a4[4], b[20];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a4[q] = seed * (threadIdx.x + q);
#pragma unroll
        for (int d = 0; d < MW; ++d) acc[q][d] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 20; ++j) b[j] = seed + j + threadIdx.x;
    for (int p = 0; p < planes; ++p) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int d0 = 0; d0 < MW; d0 += BATCH) {
                float df[BATCH];
#pragma unroll
                for (int i = 0; i < BATCH; ++i) if (d0 + i < MW) df[i] = a4[q] - b[q + d0 + i];
#pragma unroll
                for (int i = 0; i < BATCH; ++i) if (d0 + i < MW) df[i] = df[i] * df[i];
#pragma unroll
                for (int i = 0; i < BATCH; ++i) if (d0 + i < MW) acc[q][d0 + i] = acc[q][d0 + i] + df[i];
            }
        // new operands every plane (as the LDS reads deliver them): cheap, dependent on the plane index
#pragma unroll
        for (int j = 0; j < 20; j += 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(b[j]) : "v"(seed));
        asm volatile("" : "+v"(a4[0]), "+v"(a4[1]), "+v"(a4[2]), "+v"(a4[3]));
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int d = 0; d < MW; ++d) s += acc[q][d];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MW, int BATCH> void run(const char *name, int threads, int blocks_per_cu) {
    float *d; hipMalloc(&d, 256 * 8 * 1024 * 4);
    const int planes = 2000, nb = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MW, BATCH>), dim3(nb), dim3(threads), 0, 0, d, planes, 1.25f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)planes * (3.0 * 4 * MW + 4);
    const double waves_per_simd = threads / 64.0 * blocks_per_cu / 4.0;
    const double us_per_plane = ms * 1e3 / planes;
    printf("%-28s %4d threads x %d blocks/CU (%.0f waves/SIMD): %.3f ms, %.3f us per plane, %.2f ns per instruction and SIMD = %.2f cycles at 2.4 GHz (%.2f wave-instr per cycle and CU)\n",
           name, threads, blocks_per_cu, waves_per_simd, ms, us_per_plane, ms * 1e6 / (instr_per_wave * waves_per_simd), ms * 1e6 / (instr_per_wave * waves_per_simd) * 2.4,
           instr_per_wave * waves_per_simd * 4 / (ms * 1e6 * 2.4));
    hipFree(d);
}
int main() {
    run<16, 8>("16 cells, batches of 8", 1024, 1);
    run<16, 8>("16 cells, batches of 8", 512, 2);
    run<16, 8>("16 cells, batches of 8", 512, 1);
    run<16, 8>("16 cells, batches of 8", 256, 1);
    run<16, 1>("16 cells, chains", 1024, 1);
    run<16, 16>("16 cells, batches of 16", 1024, 1);
    run<17, 8>("17 cells, batches of 8", 1024, 1);
    return 0;
}
