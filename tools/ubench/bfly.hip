// What does the 8-column wave-min butterfly cost on gfx950, piece by piece?  Each variant runs ITER dependent rounds
// on 8 registers per lane; time is reported as equivalent plain-VALU instructions per round (v_add chain calibration).
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
template <int V> __global__ __launch_bounds__(1024) void k(int *out, int seed) {
    const int lane = threadIdx.x & 63;
    int key[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) key[x] = seed * (x + 3) + lane * 7 + (threadIdx.x >> 6);
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    int acc = 0;
    for (int it = 0; it < ITER; ++it) {
        if (V == 0) {   // calibration: 32 dependent-free plain adds
#pragma unroll
            for (int x = 0; x < 8; ++x) { key[x] += it; key[x] ^= lane; key[x] += 3; key[x] ^= it; }
            acc += key[0];
        } else {
            int a[4], b[2], c;
            if (V == 1 || V >= 5) {   // steps 1-2: quad_perm
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int mine = b0 ? key[2 * i + 1] : key[2 * i], other = b0 ? key[2 * i] : key[2 * i + 1];
                    a[i] = min(mine, __builtin_amdgcn_update_dpp(0, other, 0xB1, 0xf, 0xf, false));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int mine = b1 ? a[2 * i + 1] : a[2 * i], other = b1 ? a[2 * i] : a[2 * i + 1];
                    b[i] = min(mine, __builtin_amdgcn_update_dpp(0, other, 0x4E, 0xf, 0xf, false));
                }
            } else { b[0] = key[0] + key[2]; b[1] = key[1] + key[3]; }
            if (V == 2 || V >= 5) {   // steps 3-4: row_ror
                const int mine = b2 ? b[1] : b[0], other = b2 ? b[0] : b[1];
                c = min(mine, __builtin_amdgcn_update_dpp(0, other, 0x124, 0xf, 0xf, false));
                c = min(c, __builtin_amdgcn_update_dpp(0, c, 0x128, 0xf, 0xf, false));
            } else c = b[0] ^ b[1];
            if (V == 3 || V == 5) {   // steps 5-6: permlane swaps
                const auto r = __builtin_amdgcn_permlane16_swap(c, c, false, false);
                c = min((int)r[0], (int)r[1]);
                const auto q = __builtin_amdgcn_permlane32_swap(c, c, false, false);
                c = min((int)q[0], (int)q[1]);
            }
            if (V == 6) {             // steps 5-6 through the LDS crossbar instead
                c = min(c, __builtin_amdgcn_ds_swizzle(c, 0x401f));
                c = min(c, __shfl_xor(c, 32));
            }
            if (V == 4) {             // the index part: 8 x (readlane, compare, ff1, writelane)
                int wl = 0;
#define F(x) { const int f = __builtin_ctzll(__builtin_amdgcn_ballot_w64(key[x] == __builtin_amdgcn_readlane(c, x)) | 1ull << 63); \
               asm("v_writelane_b32 %0, %1, " #x : "+v"(wl) : "s"(f)); }
                F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#undef F
                c += wl;
            }
            if (V == 7) {             // index part, phased: 8 readlanes, 8 compares, 8 ff1, 8 writelanes
                int wk8[8]; unsigned long long m8[8]; int f8[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) wk8[x] = __builtin_amdgcn_readlane(c, x);
#pragma unroll
                for (int x = 0; x < 8; ++x) m8[x] = __builtin_amdgcn_ballot_w64(key[x] == wk8[x]) | 1ull << 63;
#pragma unroll
                for (int x = 0; x < 8; ++x) f8[x] = __builtin_ctzll(m8[x]);
                int wl = 0;
#define W(x) asm("v_writelane_b32 %0, %1, " #x : "+v"(wl) : "s"(f8[x]));
                W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7)
#undef W
                c += wl;
            }
            if (V == 8) {             // index part, packed: ff1 results packed on the scalar unit, one extract per lane
                unsigned long long pk = 0;
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(key[x] == __builtin_amdgcn_readlane(c, x)) | 1ull << 63;
                    pk |= (unsigned long long)__builtin_ctzll(m) << (8 * x);
                }
                c += (int)((pk >> (8 * (lane & 7))) & 0xff);
            }
            acc += c;
#pragma unroll
            for (int x = 0; x < 8; ++x) key[x] += acc;   // 8 plain ops: keeps rounds dependent
        }
    }
    if (acc == 0x7fffffff) out[0] = acc;
}
template <int V> float run(int *d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int i = 0; i < 2; ++i) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<V>, dim3(256), dim3(1024), 0, 0, d, 1);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}
int main() {
    int *d; (void)hipMalloc(&d, 4096);
    const float cal = run<0>(d) / 33.f;   // ms per plain op (x ITER x waves)
    const char *names[] = {"", "steps1-2 quad_perm (6 dpp-min + 12 cndmask)", "steps3-4 row_ror (2 dpp-min + 2 cndmask)", "steps5-6 permlane swaps",
                           "index: 8 x readlane/cmp/ff1/writelane", "whole butterfly (swaps)", "whole butterfly (swizzle+bpermute)", "index phased", "index packed on SALU"};
    float t[9];
    t[1] = run<1>(d); t[2] = run<2>(d); t[3] = run<3>(d); t[4] = run<4>(d); t[5] = run<5>(d); t[6] = run<6>(d); t[7] = run<7>(d); t[8] = run<8>(d);
    printf("plain op = %.4f ms\n", cal);
    for (int v = 1; v <= 8; ++v) printf("V%d %-48s %.1f plain-op equivalents per round (incl. ~10 ops of glue)\n", v, names[v], t[v] / cal);
    return 0;
}
