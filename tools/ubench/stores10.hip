// Which ingredient of the row-image copy-out breaks the VALU/store overlap that plain register stores enjoy
// (stores9)?  bits: 1 = 8 ds_write_b32 deposits per row before the barrier, 2 = copy data comes from LDS
// (ds_read_b128 -> store), 4 = head/tail fragment dword stores by waves 0..7, 8 = LDS tile reads (14 b128/row)
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
__global__ __launch_bounds__(1024) void k(float *out, int Ho, int Wo, int D, int TY, int bits, int spin, int nost) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *img = reinterpret_cast<float *>(smem);            // [2][8][1056]
    const float4 *tile = reinterpret_cast<const float4 *>(smem + 70 * 1024);
    const int x0 = min((int)blockIdx.x * 8, Wo - 8), y0 = min((int)blockIdx.y * TY, Ho - TY);
    float4 v = make_float4(lane, 1, 2, 3);
    float f0 = lane, f1 = lane + 1, f2 = lane + 2, f3 = lane + 3;
    for (int r = 0; r < TY; ++r) {
        if (bits & 8) {
#pragma unroll
            for (int i = 0; i < 14; ++i) { float4 t = tile[(r * 49 + i + lane) & 1023]; f0 += t.x; f1 += t.y; f2 += t.z; f3 += t.w; }
        }
        for (int s = 0; s < spin; ++s) {
            f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, 0.5f);
            f2 = __builtin_fmaf(f2, 1.0001f, 0.5f); f3 = __builtin_fmaf(f3, 1.0001f, 0.5f);
        }
        const long long G0 = ((long long)(y0 + r) * Wo + x0) * D;
        const int a0 = (int)(G0 & 31);
        float *st = img + (r & 1) * 8 * 1056;
        if (bits & 1) {
#pragma unroll
            for (int x = 0; x < 8; ++x) st[x * 1056 + ((a0 + x * D) & 31) + tid] = f0 + x;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!nost || f0 == -3.f) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int slot = tid + 1024 * k2, x = slot >> 8, j = slot & 255;
                const int ax = (a0 + x * D) & 31, head = (32 - ax) & 31;
                const int nb4 = ((1024 - head) >> 5) << 3;
                if (j < nb4) {
                    float4 val = (bits & 2) ? reinterpret_cast<const float4 *>(st + x * 1056 + ax + head)[j] : v;
                    reinterpret_cast<float4 *>(out + G0 + (long long)x * D + head)[j] = val;
                }
            }
            if ((bits & 4) && wave < 8) {
                const int x = wave;
                const int ax = (a0 + x * D) & 31, head = (32 - ax) & 31;
                const int nbody = ((1024 - head) >> 5) << 5;
                const int tail0 = head + nbody, ntail = 1024 - tail0;
                const float *sr = st + x * 1056 + ax;
                float *gr = out + G0 + (long long)x * D;
                if (lane < 32) { if (lane < head) gr[lane] = sr[lane]; }
                else if (lane - 32 < ntail) gr[tail0 + lane - 32] = sr[tail0 + lane - 32];
            }
        }
    }
    if (f0 + f1 + f2 + f3 == -1.f) out[0] = f0;
}
void run(float *d, int bits, int spin, int nost) {
    int Ho = 442, Wo = 602, D = 1089, TY = 24;
    dim3 grid((Wo + 7) / 8, (Ho + TY - 1) / TY);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, grid, dim3(1024), 115 * 1024, 0, d, Ho, Wo, D, TY, bits, spin, nost);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("bits=%2d spin=%3d nostore=%d: %.3f ms\n", bits, spin, nost, ms);
}
int main() {
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 115 * 1024);
    float *d; (void)hipMalloc(&d, 442ll * 602 * 1089 * 4 + 4096);
    for (int bits : {3, 11, 15}) for (int spin : {0, 15, 25}) { run(d, bits, spin, 0); run(d, bits, spin, 1); }
    return 0;
}
