// Store-path microbenchmark: each wave writes 64 consecutive floats (256 B) per store at a pixel stride,
// aligned or not; variants with 8/16 B per lane.  Reports TB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int VW>   // floats per lane per store
__global__ __launch_bounds__(512) void k(float *out, long long total_floats, int stride_f, int chunk_f, int nchunks, int npix_per_wave) {
    // wave w handles pixels [w*npix, (w+1)*npix); for each pixel and each chunk writes chunk_f floats
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    typedef float vt __attribute__((ext_vector_type(VW)));
    vt v;
    for (int i = 0; i < VW; ++i) v[i] = (float)lane;
    for (int p = 0; p < npix_per_wave; ++p) {
        long long pix = wave * npix_per_wave + p;
        long long base = pix * stride_f;
        if (base + stride_f > total_floats) return;
        for (int c = 0; c < nchunks; ++c) {
            long long off = base + (long long)c * chunk_f + lane * VW;
            if (c * chunk_f + lane * VW + VW <= stride_f)
                __builtin_memcpy(out + off, &v, sizeof(v));
        }
    }
}
template <int VW> void run(float *d, long long total, int stride_f, const char *name) {
    int chunk_f = 64 * VW;
    int nchunks = (stride_f + chunk_f - 1) / chunk_f;
    long long npix = total / stride_f;
    int npw = 8;
    long long nwaves = npix / npw;
    int blocks = (int)(nwaves / 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<VW>, dim3(blocks), dim3(512), 0, 0, d, total, stride_f, chunk_f, nchunks, npw);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = (double)blocks * 8 * npw * stride_f * 4;
    printf("%-28s VW=%d stride=%d floats: %.3f ms  %.2f TB/s\n", name, VW, stride_f, ms, bytes / ms / 1e9);
}
int main() {
    long long total = 290000000;   // ~1.16 GB
    float *d; hipMalloc(&d, (total + 4096) * sizeof(float));
    hipMemset(d, 0, total * 4);
    run<1>(d, total, 1089, "pixel-major 33x33 (4B align)");
    run<1>(d, total, 1088, "stride 1088 (128B align)");
    run<1>(d, total, 1024, "stride 1024");
    run<2>(d, total, 1089, "pixel-major 33x33");
    run<2>(d, total, 1088, "stride 1088");
    run<4>(d, total, 1089, "pixel-major 33x33");
    run<4>(d, total, 1088, "stride 1088");
    run<4>(d + 1, total, 1088, "stride 1088 base+4B");
    run<1>(d + 1, total, 1088, "stride 1088 base+4B");
    return 0;
}
