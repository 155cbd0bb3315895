// VALU issue rate by instruction kind on gfx950 (wave-instructions per cycle and CU), 4 waves per SIMD unless told otherwise.
// Eight independent destination registers per kind, 16 x 8 instructions per loop iteration, inline asm so the instruction is
// exactly the one named.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define BODY(INS)                                                                                                    \
    for (int i = 0; i < iters; ++i) {                                                                                \
        _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                                             \
            asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                     \
                         : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) \
                         : "v"(a), "v"(b), "s"(sa) : "vcc", "s20", "s21", "s22", "s23", "s24");                                                                 \
        }                                                                                                            \
    }
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define I_ADD(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define I_SUB(n) "v_sub_f32 %" #n ", %" #n ", %8\n"
#define I_MUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_ADDS(n) "v_add_f32 %" #n ", %10, %" #n "\n"
#define I_SUBS(n) "v_sub_f32 %" #n ", %10, %" #n "\n"
#define I_ADD3(n) "v_add_f32_e64 %" #n ", %" #n ", %8\n"
#define I_MIN(n) "v_min_i32 %" #n ", %" #n ", %8\n"
#define I_MIN3(n) "v_min3_i32 %" #n ", %" #n ", %8, %9\n"
#define I_ADDU(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_CND(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_DPP(n) "v_add_f32_dpp %" #n ", %8, %" #n " row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
#define I_SUBDPP(n) "v_sub_f32_dpp %" #n ", %8, %" #n " row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
#define I_CMP(n) "v_cmp_eq_u32 vcc, %" #n ", %8\n"
#define I_RDL(n) "v_readlane_b32 s20, %" #n ", 3\n"
#define I_MAD(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define I_LSH(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define I_CND64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\n"
#define I_CMPV(n) "v_cmp_eq_u32 vcc, %" #n ", %8\n"
#define I_CMPS(n) "v_cmp_eq_u32_e64 s[22:23], %" #n ", %8\n"
#define I_MINF(n) "v_min_f32 %" #n ", %" #n ", %8\n"
#define I_MIN3F(n) "v_min3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_ADD3U(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define I_MULU24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define I_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define I_SUBU(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define I_MINU(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define I_MOVDPP(n) "v_mov_b32_dpp %" #n ", %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
#define I_MOVS(n) "v_mov_b32 %" #n ", %10\n"
#define I_FMAS(n) "v_fma_f32 %" #n ", %" #n ", %8, %10\n"
#define I_PKADD(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define I_MIX1(n) "v_sub_f32 %" #n ", %10, %" #n "\nv_mul_f32 %" #n ", %" #n ", %8\n"
#define I_MIX2(n) "v_sub_f32 %" #n ", %10, %" #n "\nv_mul_f32 %" #n ", %" #n ", %8\nv_add_f32 %" #n ", %" #n ", %9\n"
#define I_RDL2(n) "v_readlane_b32 s24, %" #n ", 3\n"
#define I_CVT(n) "v_cvt_f16_f32 %" #n ", %" #n "\n"
#define I_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"

template <int KIND> __global__ __launch_bounds__(1024) void k(float *out, int iters, float a, float b, float sa) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j;
    if constexpr (KIND == 0) { BODY(I_FMA) }
    if constexpr (KIND == 1) { BODY(I_FMAC) }
    if constexpr (KIND == 2) { BODY(I_ADD) }
    if constexpr (KIND == 3) { BODY(I_SUB) }
    if constexpr (KIND == 4) { BODY(I_MUL) }
    if constexpr (KIND == 5) { BODY(I_ADDS) }
    if constexpr (KIND == 6) { BODY(I_ADD3) }
    if constexpr (KIND == 7) { BODY(I_MIN) }
    if constexpr (KIND == 8) { BODY(I_MIN3) }
    if constexpr (KIND == 9) { BODY(I_ADDU) }
    if constexpr (KIND == 10) { BODY(I_MOV) }
    if constexpr (KIND == 11) { BODY(I_CND) }
    if constexpr (KIND == 12) { BODY(I_DPP) }
    if constexpr (KIND == 13) { BODY(I_SUBDPP) }
    if constexpr (KIND == 14) { BODY(I_SUBS) }
    if constexpr (KIND == 15) { BODY(I_MAD) }
    if constexpr (KIND == 16) { BODY(I_LSH) }
    if constexpr (KIND == 17) { BODY(I_CND64) }
    if constexpr (KIND == 18) { BODY(I_CMPV) }
    if constexpr (KIND == 19) { BODY(I_CMPS) }
    if constexpr (KIND == 20) { BODY(I_MINF) }
    if constexpr (KIND == 21) { BODY(I_MIN3F) }
    if constexpr (KIND == 22) { BODY(I_AND) }
    if constexpr (KIND == 23) { BODY(I_ADD3U) }
    if constexpr (KIND == 24) { BODY(I_MULU24) }
    if constexpr (KIND == 25) { BODY(I_MULLO) }
    if constexpr (KIND == 26) { BODY(I_SUBU) }
    if constexpr (KIND == 27) { BODY(I_MINU) }
    if constexpr (KIND == 28) { BODY(I_MOVDPP) }
    if constexpr (KIND == 29) { BODY(I_MOVS) }
    if constexpr (KIND == 30) { BODY(I_FMAS) }
    if constexpr (KIND == 31) { BODY(I_MIX1) }
    if constexpr (KIND == 32) { BODY(I_MIX2) }
    if constexpr (KIND == 33) { BODY(I_RDL2) }
    if constexpr (KIND == 34) { BODY(I_CVT) }
    if constexpr (KIND == 35) { BODY(I_XOR) }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static const char *names[] = {"v_fma_f32 vvv", "v_fmac_f32", "v_add_f32 vv", "v_sub_f32 vv", "v_mul_f32 vv", "v_add_f32 sv", "v_add_f32_e64", "v_min_i32",
                              "v_min3_i32", "v_add_u32", "v_mov_b32", "v_cndmask vcc", "v_add_f32 dpp", "v_sub_f32 dpp", "v_sub_f32 sv", "v_mad_u32_u24", "v_lshlrev_b32", "v_cndmask s[]", "v_cmp vcc", "v_cmp_e64 s[]", "v_min_f32", "v_min3_f32", "v_and_b32", "v_add3_u32", "v_mul_u32_u24", "v_mul_lo_u32", "v_sub_u32", "v_min_u32", "v_mov_dpp", "v_mov v,s", "v_fma vvs", "mix sub_sv+mul (x2)", "mix sub_sv+mul+add (x3)", "v_readlane", "v_cvt_f16_f32", "v_xor_b32"};
static const int per[] = {1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1, 1,1,1,1,1,1,1,1,1,1,1,1,1,1,2,3,1,1,1};
template <int KIND> void run(float *d, int wps, int blocks_per_cu) {
    int iters = 2048;
    int blocks = 256 * blocks_per_cu, threads = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 64, 1.0001f, 0.5f, 0.25f);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f, 0.25f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double winstr = (double)blocks * (threads / 64) * iters * 128.0 * per[KIND];
    double per_cu_per_us = winstr / 256 / (best * 1e3);
    printf("%-16s waves/SIMD=%d  %.3f ms  %.0f wave-instr/us/CU (%.2f per cycle @2.4GHz)\n", names[KIND], wps * blocks_per_cu, best, per_cu_per_us, per_cu_per_us / 2400.0);
}
template <int KIND> void both(float *d) { run<KIND>(d, 4, 1); run<KIND>(d, 1, 1); }
int main() {
    float *d; hipMalloc(&d, 256 * 2 * 1024 * sizeof(float));
    if (getenv("VALU_FIRST")) {
    both<0>(d); both<1>(d); both<2>(d); both<3>(d); both<4>(d); both<5>(d); both<6>(d); both<7>(d); both<8>(d); both<9>(d); both<10>(d);
    both<11>(d); both<12>(d); both<13>(d); both<14>(d); both<15>(d); both<16>(d);
    run<0>(d, 2, 1); run<2>(d, 2, 1); run<0>(d, 4, 2); run<2>(d, 4, 2);
    }
    run<2>(d, 4, 1); run<11>(d, 4, 1); run<17>(d, 4, 1); run<18>(d, 4, 1); run<19>(d, 4, 1); run<20>(d, 4, 1); run<21>(d, 4, 1); run<22>(d, 4, 1); run<23>(d, 4, 1);
    run<24>(d, 4, 1); run<25>(d, 4, 1); run<26>(d, 4, 1); run<27>(d, 4, 1); run<28>(d, 4, 1); run<29>(d, 4, 1); run<30>(d, 4, 1); run<31>(d, 4, 1); run<32>(d, 4, 1);
    run<33>(d, 4, 1); run<34>(d, 4, 1); run<35>(d, 4, 1);
    return 0;
}
