// Which write-front structure does the memory system like?  Persistent blocks (one per CU) write the cost volume's 8-pixel
// runs (34 848 B, whole 128-B lines, non-temporal dwordx4) row by row; the schedule decides how many distinct fronts exist:
//   g = tiles per group: g adjacent CUs walk g adjacent tile columns in lockstep (a front of g * 34 KB contiguous bytes);
//   the (super-column, row) sequence is cut evenly over the B/g groups (g = 1: the column-major persistent sweep).
//   g = 0: the static-tile grid (76 x 10 blocks of 48 rows, XCD-aware order) for reference.
// `spin` FMAs per thread and row mimic the compute that paces the real kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_run(float *out, long long G0, int RUN, int tid, f4_t v) {
    const long long A = G0 & ~31ll;                          // line-aligned start
    const int n4 = (int)(((G0 + RUN + 31) & ~31ll) - A) >> 2;  // float4 pieces
    const float *gb = out + A;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (tid + i * 1024 < n4)
            asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"((unsigned)(tid + i * 1024) * 16u), "v"(v), "s"(gb) : "memory");
}
__global__ __launch_bounds__(1024) void k(float *out, int Ho, int Wo, int D, int g, int spin, int ty) {
    const int tid = threadIdx.x;
    const int ncols = (Wo + 7) / 8, RUN = 8 * D;
    f4_t v = {(float)tid, 1, 2, 3};
    float f0 = tid;
    if (g <= -1000) {            // row-major persistent: tile t = y * ncols + c, block b takes t = b, b + B, ...
        const int B = gridDim.x;
        for (int t = blockIdx.x; t < ncols * Ho; t += B) {
            const int y = t / ncols, c = t - y * ncols;
            const int x0 = min(c * 8, Wo - 8);
            for (int s = 0; s < spin; ++s) f0 = __builtin_fmaf(f0, 1.0001f, 0.5f);
            asm volatile("s_barrier" ::: "memory");
            store_run(out, ((long long)y * Wo + x0) * D, RUN, tid, v);
        }
    } else if (g == 0) {
        int bx, by;
        const int nb = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
        const int per = nb >> 3, rem = nb & 7, xcd = lin & 7, slot = lin >> 3;
        const int t = xcd * per + min(xcd, rem) + slot;
        by = t / (int)gridDim.x; bx = t - by * (int)gridDim.x;
        const int x0 = min(bx * 8, Wo - 8), y0 = min(by * ty, Ho - ty);
        for (int r = 0; r < ty + 6; ++r) {
            for (int s = 0; s < spin; ++s) f0 = __builtin_fmaf(f0, 1.0001f, 0.5f);
            asm volatile("s_barrier" ::: "memory");
            if (r >= 6) store_run(out, ((long long)(y0 + r - 6) * Wo + x0) * D, RUN, tid, v);
        }
    } else {
        const int NG = gridDim.x / g, G = blockIdx.x / g, m = blockIdx.x % g;
        if (G >= NG) return;
        const int nsc = (ncols + g - 1) / g;
        const long long tot = (long long)nsc * Ho;
        long long pos = tot * G / NG, pend = tot * (G + 1) / NG;
        while (pos < pend) {
            const int sc = (int)(pos / Ho), row = (int)(pos - (long long)sc * Ho);
            const int n = (int)min((long long)(Ho - row), pend - pos);
            const int col = sc * g + m;
            const int x0 = min(col * 8, Wo - 8);
            for (int r = 0; r < n + 6; ++r) {
                for (int s = 0; s < spin; ++s) f0 = __builtin_fmaf(f0, 1.0001f, 0.5f);
                asm volatile("s_barrier" ::: "memory");
                if (r >= 6 && col < ncols) store_run(out, ((long long)(row + r - 6) * Wo + x0) * D, RUN, tid, v);
            }
            pos += n;
        }
    }
    if (f0 == -1.f) out[0] = f0;
}
int main(int argc, char **argv) {
    const int Ho = 442, Wo = 602, D = 1089;
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    float *d;
    (void)hipMalloc(&d, (size_t)Ho * Wo * D * 4 + 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double bytes = (double)Ho * Wo * D * 4;
    for (int rep = 0; rep < 2; ++rep)
    for (int spin : {0, 40}) {
        for (int g : {0, -2, -3, -4, -5, 1, 1228, 1200, 1160, 1128, -1256, -1228, -1192, -1128}) {
            float best = 1e9;
            for (int it = 0; it < 6; ++it) {
                (void)hipEventRecord(e0);
                if (g == 0) hipLaunchKernelGGL(k, dim3(76, 10), dim3(1024), 100 * 1024, 0, d, Ho, Wo, D, 0, spin, 48);
                else if (g <= -1000) hipLaunchKernelGGL(k, dim3(-g - 1000), dim3(1024), 100 * 1024, 0, d, Ho, Wo, D, g, spin, 0);
                else if (g < 0) hipLaunchKernelGGL(k, dim3(76, -g), dim3(1024), 100 * 1024, 0, d, Ho, Wo, D, 0, spin, (Ho - g - 1) / -g);
                else if (g > 1000) hipLaunchKernelGGL(k, dim3(g - 1000), dim3(1024), 100 * 1024, 0, d, Ho, Wo, D, 1, spin, 0);
                else hipLaunchKernelGGL(k, dim3(256), dim3(1024), 100 * 1024, 0, d, Ho, Wo, D, g, spin, 0);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (it >= 2 && ms < best) best = ms;
            }
            printf("spin=%3d g=%2d: %.1f us  %.2f TB/s\n", spin, g, best * 1e3, bytes / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
