// Rotating single-wave copier: per row ONE wave of the block writes the whole row image (8 runs of RUNF floats,
// line-aligned dwordx4 bursts of 1 KB, head/tail fragments skipped), the role rotating over the NW waves.
// WPB waves per block participate as copiers (others idle); NCOP copiers per row split the runs.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int RUNF, int NW, int NCOP>
__global__ __launch_bounds__(NW * 64) void k(float *out, int Ho, int Wo, int D, int TY) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0 = min((int)blockIdx.x * 8, Wo - 8), y0 = min((int)blockIdx.y * TY, Ho - TY);
    float4 v = make_float4(lane, 1, 2, 3);
    for (int r = 0; r < TY; ++r) {
        int role = (wave - r * NCOP) % NW; if (role < 0) role += NW;
        if (role < NCOP) {
            for (int x = role; x < 8; x += NCOP) {
                long long G = ((long long)(y0 + r) * Wo + x0 + x) * D;
                int head = (int)((32 - (G & 31)) & 31);
                int nb4 = ((RUNF - head) >> 5) << 3;
                float4 *gb = reinterpret_cast<float4 *>(out + G + head);
                for (int j = lane; j < nb4; j += 64) gb[j] = v;
            }
        }
    }
}
template <int RUNF, int NW, int NCOP> void run(float *d, int TY) {
    int Ho = 442, Wo = 602, D = 1089;
    dim3 grid((Wo + 7) / 8, (Ho + TY - 1) / TY);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<RUNF, NW, NCOP>), grid, dim3(NW * 64), 0, 0, d, Ho, Wo, D, TY);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("runs of %4d floats, %2d waves/block, %d copier(s) per row, TY=%d: %.3f ms  %.2f TB/s\n", RUNF, NW, NCOP, TY, ms,
           442.0 * 602 * RUNF * 4 / ms / 1e9);
}
int main() {
    float *d; (void)hipMalloc(&d, 442ll * 602 * 1089 * 4 + 4096);
    run<1024, 16, 1>(d, 24); run<1024, 16, 2>(d, 24); run<1024, 16, 4>(d, 24); run<1088, 16, 1>(d, 24); run<1024, 4, 1>(d, 24); run<1024, 1, 1>(d, 24);
    return 0;
}
