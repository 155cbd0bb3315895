// The arithmetic of one row-image task row (14 squared differences with frame-0 operands in SGPRs, block-form horizontal
// sums, pair-sum vertical ring) on register-resident data: what VALU issue rate does this instruction mix reach by itself?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef const float __attribute__((address_space(4))) *cfptr;
extern __shared__ float4 lds4[];
template <int VAR> __global__ __launch_bounds__(1024) void k(const float *__restrict__ a0, float *out, int iters) {
    if (VAR >= 3) {
        for (int i = threadIdx.x; i < 64 * 49; i += blockDim.x) lds4[i] = make_float4(i * 0.5f, i * 0.25f, i * 0.125f, 0.f);
        __syncthreads();
    }
    float *stage = reinterpret_cast<float *>(lds4 + 64 * 49);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 *lr0 = lds4 + (lane / 33) * 49 + (lane % 33);
    float ring[6][8];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int x = 0; x < 8; ++x) ring[i][x] = 0.f;
    float av[3][14];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int s = 0; s < 14; ++s) av[c][s] = ((cfptr)a0)[c * 14 + s];    // uniform -> SGPRs
    float bx = threadIdx.x * 0.25f, acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            float e[14];
            float4 bb[14];
            if (VAR >= 3) {
                const float4 *lr = lr0 + ((it * 6 + m) & 15) * 49;
#pragma unroll
                for (int s = 0; s < 5; ++s) bb[s] = lr[s];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < 14; ++s) {
                if (VAR >= 3 && (s == 5 || s == 10)) {
                    const float4 *lr = lr0 + ((it * 6 + m) & 15) * 49;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = s; t < s + 5 && t < 14; ++t) bb[t] = lr[t];
                    __builtin_amdgcn_sched_barrier(0);
                }
                float b0 = bx + s, b1 = bx - s, b2 = bx * 0.5f;   // stand-ins for the LDS pixels (3 cheap VALU, counted below)
                if (VAR >= 3) { b0 = bb[s].x; b1 = bb[s].y; b2 = bb[s].z; asm volatile("" :: "v"(bb[s].w)); }
                if (VAR == 1) { asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2)); }
                float d0, d1, d2;
                if (VAR == 2) {   // frame-0 operands from VGPRs instead of SGPRs
                    float a0v = av[0][s], a1v = av[1][s], a2v = av[2][s];
                    asm volatile("" : "+v"(a0v), "+v"(a1v), "+v"(a2v));
                    d0 = a0v - b0; d1 = a1v - b1; d2 = a2v - b2;
                } else {
                    d0 = av[0][s] - b0; d1 = av[1][s] - b1; d2 = av[2][s] - b2;
                }
                e[s] = fmaf(d2, d2, fmaf(d1, d1, d0 * d0));
            }
            float sa[7], pb[7], h[8];
            sa[6] = e[6];
#pragma unroll
            for (int i = 5; i >= 0; --i) sa[i] = e[i] + sa[i + 1];
            pb[0] = e[7];
#pragma unroll
            for (int j = 1; j < 7; ++j) pb[j] = pb[j - 1] + e[7 + j];
            h[0] = sa[0];
#pragma unroll
            for (int x = 1; x < 7; ++x) h[x] = sa[x] + pb[x - 1];
            h[7] = pb[6];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                float v = (ring[m][x] + ring[(m + 2) % 6][x]) + (ring[(m + 4) % 6][x] + h[x]);
                ring[(m + 5) % 6][x] += h[x];
                ring[m][x] = h[x];
                if (VAR >= 4) stage[(m & 1) * 8800 + x * 1089 + wave * 64 + lane] = v; else
                acc += v;                       // stands in for the deposit
            }
            if (VAR >= 5) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            bx += 1.f;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// The same task row with the frame-1 tile PLANAR in LDS ([row][channel][75] floats: row stride 225 == 33 mod 32, conflict-free) and
// the squared differences in PACKED fp32 (two pixels per instruction): per channel 7 ds_read2_b32 give the pixel pairs (2j, 2j+1),
// v_pk_add_f32 with the frame-0 pair as an SGPR-pair operand and neg modifiers subtracts, v_pk_mul / v_pk_fma square and add.  21 LDS
// reads of 512 B instead of 14 of 1 KB; 42 packed instead of 84 plain VALU instructions for the 14 squared differences.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef f2 f2u __attribute__((aligned(4)));
extern __shared__ float ldsf[];
template <int VAR, bool HYB = false> __global__ __launch_bounds__(1024) void kp(const float *__restrict__ a0, float *out, int iters) {
    constexpr int PP = 75, RS = 3 * PP;
    for (int i = threadIdx.x; i < 64 * RS; i += blockDim.x) ldsf[i] = i * 0.5f;
    __syncthreads();
    float *stage = ldsf + 64 * RS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *lr0 = ldsf + (lane / 33) * RS + (lane % 33);
    float ring[6][8];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int x = 0; x < 8; ++x) ring[i][x] = 0.f;
    f2 av[3][7];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int s = 0; s < 7; ++s) av[c][s] = f2{((cfptr)a0)[c * 14 + 2 * s], ((cfptr)a0)[c * 14 + 2 * s + 1]};    // uniform -> SGPR pairs
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const float *lr = lr0 + ((it * 6 + m) & 15) * RS;
            f2 e2[7];
            float eh[14];       // HYB: only the subtraction is packed (21 instead of 42 slow-path issues); the squares stay plain fast-path ops
            f2 b[2][7];
#pragma unroll
            for (int j = 0; j < 7; ++j) b[0][j] = *reinterpret_cast<const f2u *>(lr + 2 * j);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                __builtin_amdgcn_sched_barrier(0);
                if (c < 2) {
#pragma unroll
                    for (int j = 0; j < 7; ++j) b[(c + 1) & 1][j] = *reinterpret_cast<const f2u *>(lr + (c + 1) * PP + 2 * j);
                }
                __builtin_amdgcn_sched_barrier(0);
                // (all seven differences first, in place over the pixel pairs, then the seven squares: a packed op that consumes the
                //  result of the one just in front of it costs a wait state and the full pipeline latency)
#pragma unroll
                for (int j = 0; j < 7; ++j) asm("v_pk_add_f32 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(b[c & 1][j]) : "s"(av[c][j]));
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    if constexpr (HYB) {
                        const float d0 = b[c & 1][j][0], d1 = b[c & 1][j][1];
                        if (c == 0) { asm("v_mul_f32 %0, %1, %1" : "=v"(eh[2 * j]) : "v"(d0)); asm("v_mul_f32 %0, %1, %1" : "=v"(eh[2 * j + 1]) : "v"(d1)); }   // (asm: the compiler would re-pack the pair)
                        else { eh[2 * j] = fmaf(d0, d0, eh[2 * j]); eh[2 * j + 1] = fmaf(d1, d1, eh[2 * j + 1]); }
                    } else {
                        if (c == 0) asm("v_pk_mul_f32 %0, %1, %1" : "=v"(e2[j]) : "v"(b[c & 1][j]));
                        else asm("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(e2[j]) : "v"(b[c & 1][j]));
                    }
                }
            }
            float e[14];
#pragma unroll
            for (int j = 0; j < 7; ++j) { e[2 * j] = HYB ? eh[2 * j] : e2[j][0]; e[2 * j + 1] = HYB ? eh[2 * j + 1] : e2[j][1]; }
            float sa[7], pb[7], h[8];
            sa[6] = e[6];
#pragma unroll
            for (int i = 5; i >= 0; --i) sa[i] = e[i] + sa[i + 1];
            pb[0] = e[7];
#pragma unroll
            for (int j = 1; j < 7; ++j) pb[j] = pb[j - 1] + e[7 + j];
            h[0] = sa[0];
#pragma unroll
            for (int x = 1; x < 7; ++x) h[x] = sa[x] + pb[x - 1];
            h[7] = pb[6];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                float v = (ring[m][x] + ring[(m + 2) % 6][x]) + (ring[(m + 4) % 6][x] + h[x]);
                ring[(m + 5) % 6][x] += h[x];
                ring[m][x] = h[x];
                if (VAR >= 1) stage[(m & 1) * 8800 + x * 1089 + wave * 64 + lane] = v; else
                acc += v;
            }
            if (VAR >= 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// Round 5: the same task row on BYTE-valued frames in integer arithmetic (verdict r4 item 4): a pixel is two dwords -- (r | g << 16)
// and (b) as packed 16-bit lanes -- so the frame-1 tile costs 8 B per pixel in LDS instead of 16 (ds_read_b64), the three channel
// differences are two v_pk_sub_i16 (frame-0 operands in SGPRs), their squares and the sum two v_dot2c_i32_i16 (exact: 147 * 255^2
// < 2^31), the sliding sums v_add_u32; the eight outputs of a row are converted to fp32 (v_cvt_f32_i32) where they are deposited.
// 4 instead of 6 instructions per position: 56 + 8 instead of 84 of the row's ~126.
typedef short s2_t __attribute__((ext_vector_type(2)));
extern __shared__ int2 ldsi2[];
template <int VAR> __global__ __launch_bounds__(1024) void ki(const float *__restrict__ a0, float *out, int iters) {
    for (int i = threadIdx.x; i < 64 * 49; i += blockDim.x) ldsi2[i] = make_int2((i & 255) | ((i * 7 & 255) << 16), (i * 13) & 255);
    __syncthreads();
    float *stage = reinterpret_cast<float *>(ldsi2 + 64 * 49);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int2 *lr0 = ldsi2 + (lane / 33) * 49 + (lane % 33);
    int ring[6][8];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int x = 0; x < 8; ++x) ring[i][x] = 0;
    int av[2][14];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int s = 0; s < 14; ++s) av[c][s] = ((const int __attribute__((address_space(4))) *)a0)[c * 14 + s];    // uniform -> SGPRs
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            int e[14];
            int2 bb[14];
            const int2 *lr = lr0 + ((it * 6 + m) & 15) * 49;
#pragma unroll
            for (int s = 0; s < 7; ++s) bb[s] = lr[s];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 14; ++s) {
                if (s == 7) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 7; t < 14; ++t) bb[t] = lr[t];
                    __builtin_amdgcn_sched_barrier(0);
                }
                const s2_t d01 = __builtin_bit_cast(s2_t, av[0][s]) - __builtin_bit_cast(s2_t, bb[s].x);
                const s2_t d2 = __builtin_bit_cast(s2_t, av[1][s]) - __builtin_bit_cast(s2_t, bb[s].y);
                e[s] = __builtin_amdgcn_sdot2(d2, d2, __builtin_amdgcn_sdot2(d01, d01, 0, false), false);
            }
            int sa[7], pb[7], h[8];
            sa[6] = e[6];
#pragma unroll
            for (int i = 5; i >= 0; --i) sa[i] = e[i] + sa[i + 1];
            pb[0] = e[7];
#pragma unroll
            for (int j = 1; j < 7; ++j) pb[j] = pb[j - 1] + e[7 + j];
            h[0] = sa[0];
#pragma unroll
            for (int x = 1; x < 7; ++x) h[x] = sa[x] + pb[x - 1];
            h[7] = pb[6];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int v = (ring[m][x] + ring[(m + 2) % 6][x]) + (ring[(m + 4) % 6][x] + h[x]);
                ring[(m + 5) % 6][x] += h[x];
                ring[m][x] = h[x];
                if (VAR >= 1) stage[(m & 1) * 8800 + x * 1089 + wave * 64 + lane] = (float)v; else
                acc += (float)v;
            }
            if (VAR >= 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int VAR> void runi(const float *a0, float *d, int wps, const char *name) {
    int iters = 512, blocks = 256, threads = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void *)ki<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
    hipLaunchKernelGGL((ki<VAR>), dim3(blocks), dim3(threads), 140000, 0, a0, d, 8);
    (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((ki<VAR>), dim3(blocks), dim3(threads), 140000, 0, a0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double rows = (double)iters * 6;
    double us_per_row = best * 1e3 / rows;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.4f us per row step (all waves of a SIMD)  = %.0f cycles @2.3GHz per wave-row\n", name, wps, best,
           us_per_row, us_per_row * 2300.0 / wps);
}
template <int VAR, bool HYB = false> void runp(const float *a0, float *d, int wps, const char *name) {
    int iters = 512, blocks = 256, threads = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void *)kp<VAR, HYB>, hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
    hipLaunchKernelGGL((kp<VAR, HYB>), dim3(blocks), dim3(threads), 140000, 0, a0, d, 8);
    (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((kp<VAR, HYB>), dim3(blocks), dim3(threads), 140000, 0, a0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double rows = (double)iters * 6;
    double us_per_row = best * 1e3 / rows;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.4f us per row step (all waves of a SIMD)  = %.0f cycles @2.3GHz per wave-row\n", name, wps, best,
           us_per_row, us_per_row * 2300.0 / wps);
}
template <int VAR> void run(const float *a0, float *d, int wps, const char *name) {
    int iters = 512, blocks = 256, threads = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void *)k<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
    hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(threads), 140000, 0, a0, d, 8);
    (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(threads), 140000, 0, a0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double rows = (double)iters * 6;                         // task rows per wave
    double us_per_row = best * 1e3 / rows;                   // per wave-row with wps waves on each SIMD -> per SIMD: x wps rows
    printf("%-28s waves/SIMD=%d  %.3f ms  %.4f us per row step (all waves of a SIMD)  = %.0f cycles @2.3GHz per wave-row\n", name, wps, best,
           us_per_row, us_per_row * 2300.0 / wps);
}
int main() {
    float *a0, *d; (void)hipMalloc(&a0, 64 * 4); (void)hipMalloc(&d, 256 * 1024 * 4);
    (void)hipMemset(a0, 0, 64 * 4);
    for (int w : {4, 2, 1}) { run<0>(a0, d, w, "arithmetic only"); run<3>(a0, d, w, "+ 14 ds_read_b128"); run<4>(a0, d, w, "+ reads + 8 ds_write_b32"); run<5>(a0, d, w, "+ reads, writes, barrier"); }
    for (int w : {4, 2, 1}) { runi<0>(a0, d, w, "INT16 dot2: reads + arith"); runi<1>(a0, d, w, "INT16 + 8 cvt + ds_write_b32"); runi<2>(a0, d, w, "INT16 + writes, barrier"); }
    for (int w : {4, 2, 1}) { runp<0>(a0, d, w, "PACKED planar: reads + arith"); runp<1>(a0, d, w, "PACKED + 8 ds_write_b32"); runp<2>(a0, d, w, "PACKED + writes, barrier"); }
    for (int w : {4, 2, 1}) { runp<0, true>(a0, d, w, "HYBRID (pk sub only): reads + arith"); runp<1, true>(a0, d, w, "HYBRID + 8 ds_write_b32"); runp<2, true>(a0, d, w, "HYBRID + writes, barrier"); }
    return 0;
}
