// Order-A store pattern (a wave writes one piece per pixel at pixel stride, neighbours written by other waves
// at other times) with wider pieces: VW floats per lane (piece = 64*VW floats), TX pixels per task-row.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int VW> struct V { typedef float t __attribute__((ext_vector_type(VW))); };
template <> struct V<1> { typedef float t; };
template <int VW, int TX>
__global__ __launch_bounds__(512) void k(float *out, int Ho, int Wo, int D, int TY) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int piece_f = 64 * VW;
    const int npc = D / piece_f;                 // full pieces per pixel (tail ignored)
    const int GX = 32;
    const int x0 = min((int)blockIdx.x * GX, Wo - GX), y0 = min((int)blockIdx.y * TY, Ho - TY);
    typename V<VW>::t v;
    if constexpr (VW == 1) v = lane; else for (int i = 0; i < VW; ++i) v[i] = lane;
    const int ntile = GX / TX;
    for (int t = wave; t < ntile * npc; t += 8) {
        int tile = t / npc, pc = t - tile * npc;
        for (int r = 0; r < TY; ++r) {
            float *orow = out + ((long long)(y0 + r) * Wo + x0 + tile * TX) * D + pc * piece_f + lane * VW;
#pragma unroll
            for (int x = 0; x < TX; ++x) __builtin_memcpy(orow + (long long)x * D, &v, sizeof(v));
        }
    }
}
template <int VW, int TX> void run(float *d) {
    int Ho = 442, Wo = 602, D = 1089, TY = 18;
    dim3 grid((Wo + 31) / 32, (Ho + TY - 1) / TY);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<VW, TX>), grid, dim3(512), 0, 0, d, Ho, Wo, D, TY);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    double bytes = 442.0 * 602 * (D / (64 * VW)) * 64 * VW * 4;
    printf("order A  VW=%d TX=%d: %.3f ms  %.2f TB/s\n", VW, TX, ms, bytes / ms / 1e9);
}
int main() {
    float *d; (void)hipMalloc(&d, 442ll * 602 * 1089 * 4 + 4096);
    run<1, 8>(d); run<2, 8>(d); run<2, 4>(d); run<4, 8>(d); run<4, 4>(d); run<4, 2>(d);
    return 0;
}
