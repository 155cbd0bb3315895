// Packed-fp32 VALU issue rate on gfx950 (wave-instructions per cycle and CU; one instruction = 2 fp32 results per lane):
// v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 on register pairs, with an SGPR-pair operand, with op_sel swaps and neg
// modifiers -- the forms the packed sweep of the cost-volume kernel needs -- next to their scalar-lane counterparts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define BODY(INS)                                                                                                    \
    for (int i = 0; i < iters; ++i) {                                                                                \
        _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                                             \
            asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                     \
                         : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) \
                         : "v"(a), "v"(b), "s"(sa));                                                                 \
        }                                                                                                            \
    }
#define I_PKADD(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define I_PKMUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define I_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PKADDS(n) "v_pk_add_f32 %" #n ", %10, %" #n "\n"
#define I_PKSUBS(n) "v_pk_add_f32 %" #n ", %10, %" #n " neg_lo:[0,1] neg_hi:[0,1]\n"
#define I_PKADDSW(n) "v_pk_add_f32 %" #n ", %" #n ", %8 op_sel:[0,1] op_sel_hi:[1,0]\n"
#define I_PKFMASQ(n) "v_pk_fma_f32 %" #n ", %8, %8, %" #n "\n"
#define I_PKMOV(n) "v_pk_mov_b32 %" #n ", %8, %9 op_sel:[1,0]\n"
template <int KIND> __global__ __launch_bounds__(1024) void k(float *out, int iters, f2 a, f2 b, f2 sa) {
    f2 x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = f2{(float)threadIdx.x + j, (float)j};
    if constexpr (KIND == 0) { BODY(I_PKADD) }
    if constexpr (KIND == 1) { BODY(I_PKMUL) }
    if constexpr (KIND == 2) { BODY(I_PKFMA) }
    if constexpr (KIND == 3) { BODY(I_PKADDS) }
    if constexpr (KIND == 4) { BODY(I_PKSUBS) }
    if constexpr (KIND == 5) { BODY(I_PKADDSW) }
    if constexpr (KIND == 6) { BODY(I_PKFMASQ) }
    if constexpr (KIND == 7) { BODY(I_PKMOV) }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j][0] + x[j][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the squared-difference chain of one pixel pair as the packed sweep would issue it: 3 pk_sub (SGPR pair - VGPR pair), pk_mul, 2 pk_fma
__global__ __launch_bounds__(1024) void kmix(float *out, int iters, f2 b0, f2 b1, f2 b2, f2 s0, f2 s1, f2 s2) {
    f2 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f2{(float)threadIdx.x, (float)j};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f2 d0, d1, d2, e;
                asm volatile("v_pk_add_f32 %0, %4, %7 neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %1, %5, %8 neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_add_f32 %2, %6, %9 neg_lo:[0,1] neg_hi:[0,1]\n"
                             "v_pk_mul_f32 %3, %0, %0\n"
                             "v_pk_fma_f32 %3, %1, %1, %3\n"
                             "v_pk_fma_f32 %3, %2, %2, %3\n"
                             : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(e)
                             : "s"(s0), "s"(s1), "s"(s2), "v"(b0 + acc[j]), "v"(b1), "v"(b2));
                acc[j] = e;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][0] + acc[3][1];
}
static const char *names[] = {"v_pk_add_f32 vv", "v_pk_mul_f32 vv", "v_pk_fma_f32 vvv", "v_pk_add_f32 sv", "v_pk_add_f32 s,-v", "v_pk_add_f32 op_sel swap", "v_pk_fma_f32 a,a,acc", "v_pk_mov_b32"};
template <int KIND> void run(float *d, int wps) {
    int iters = 2048, blocks = 256, threads = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f2 a{1.0001f, 0.9999f}, b{0.5f, 0.25f}, sa{0.25f, 0.125f};
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 64, a, b, sa);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, a, b, sa);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double winstr = (double)blocks * (threads / 64) * iters * 128.0;
    double per_cu_per_us = winstr / 256 / (best * 1e3);
    printf("%-26s waves/SIMD=%d  %.3f ms  %.0f wave-instr/us/CU (%.2f per cycle @2.4GHz)\n", names[KIND], wps, best, per_cu_per_us, per_cu_per_us / 2400.0);
}
int main() {
    float *d; hipMalloc(&d, 256 * 1024 * sizeof(float));
    run<0>(d, 4); run<1>(d, 4); run<2>(d, 4); run<3>(d, 4); run<4>(d, 4); run<5>(d, 4); run<6>(d, 4); run<7>(d, 4);
    run<0>(d, 1); run<2>(d, 1); run<0>(d, 2);
    {
        int iters = 1024, blocks = 256, threads = 1024;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        f2 b{1.f, 2.f}, s{0.5f, 0.75f};
        hipLaunchKernelGGL(kmix, dim3(blocks), dim3(threads), 0, 0, d, 8, b, b, b, s, s, s);
        hipDeviceSynchronize();
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kmix, dim3(blocks), dim3(threads), 0, 0, d, iters, b, b, b, s, s, s);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        double winstr = (double)blocks * 16 * iters * 8 * 4 * 7.0;   // (6 packed ops + the compiler's pk add of acc)
        printf("packed sqdiff chain (7 pk ops per pixel pair)  %.3f ms  %.2f wave-instr per cycle and CU @2.4GHz\n", best, winstr / 256 / (best * 1e3) / 2400.0);
    }
    return 0;
}
