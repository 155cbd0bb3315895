// Tuning only: does physically contiguous device memory (hipExtMallocWithFlags + hipDeviceMallocContiguous) behave differently from a plain
// hipMalloc?  Streaming read / write, a read that fits the L2 / the Infinity Cache (re-read), and 10 planes read side by side, on both.
// build: hipcc -O3 --offload-arch=gfx950 contig.hip -o contig
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_write(float4 *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void k_read(const float4 *p, size_t n, float *sink) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.f) *sink = s;
}
// 10 planes of `plane` float4 each, every thread reads the same offset of all of them (a convolution's input planes)
__global__ void k_planes(const float4 *p, size_t plane, int np, float *sink) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < plane; i += (size_t)gridDim.x * blockDim.x)
        for (int c = 0; c < np; ++c) { float4 v = p[c * plane + i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.f) *sink = s;
}
// tiles: block b reads a 132 x 12 float tile of each of np planes (pitch W) and writes 128 x 8 of nout planes: the batched convolution's traffic
__global__ void k_tiles(const float *in, float *out, int H, int W, int np, int nout, float *sink) {
    const int tx = blockIdx.x % (W / 128), ty = blockIdx.x / (W / 128);
    if (ty * 8 + 12 > H) return;
    float s = 0.f;
    for (int c = 0; c < np; ++c)
        for (int i = threadIdx.x; i < 132 * 12; i += blockDim.x) {
            int y = ty * 8 + i / 132, x = tx * 128 + i % 132;
            if (x < W) s += in[((size_t)c * H + y) * W + x];
        }
    for (int c = 0; c < nout; ++c)
        for (int i = threadIdx.x; i < 128 * 8; i += blockDim.x) out[((size_t)c * H + ty * 8 + i / 128) * W + tx * 128 + i % 128] = s;
    if (s == 12345.f) *sink = s;
}
template <class F> float timeit(F f, int n = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a)); for (int i = 0; i < n; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / n;
}
int main() {
    float *sink; CK(hipMalloc(&sink, 4));
    for (int rep = 0; rep < 2; ++rep)
    for (int contig = 0; contig < 2; ++contig) {
        const size_t big = (size_t)1200 << 20, small = (size_t)48 << 20;
        void *pb, *ps;
        if (contig) { CK(hipExtMallocWithFlags(&pb, big, hipDeviceMallocContiguous)); CK(hipExtMallocWithFlags(&ps, small, hipDeviceMallocContiguous)); }
        else { CK(hipMalloc(&pb, big)); CK(hipMalloc(&ps, small)); }
        float w = timeit([&] { hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, (float4 *)pb, big / 16); });
        float r = timeit([&] { hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, (const float4 *)pb, big / 16, sink); });
        float rs = timeit([&] { hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, (const float4 *)ps, small / 16, sink); }, 100);
        float ws = timeit([&] { hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, (float4 *)ps, small / 16); }, 100);
        const size_t plane = (size_t)472 * 632 * 4 / 16;
        float pl = timeit([&] { hipLaunchKernelGGL(k_planes, dim3(2048), dim3(256), 0, 0, (const float4 *)ps, plane, 10, sink); }, 100);
        const int H = 480, W = 640;
        float *tin = (float *)ps, *tout = (float *)ps + (size_t)10 * H * W;
        float tl = timeit([&] { hipLaunchKernelGGL(k_tiles, dim3((W / 128) * (H / 8)), dim3(256), 0, 0, tin, tout, H, W, 10, 10, sink); }, 100);
        printf("%-10s %p/%p  write 1.2GB %.1f GB/s  read 1.2GB %.1f GB/s | 48MB: read %.1f GB/s write %.1f GB/s | 10 planes read %.1f GB/s (%.1f us) | conv-like tiles %.1f us\n",
               contig ? "contiguous" : "plain", pb, ps, big / w * 1e-6, big / r * 1e-6, small / rs * 1e-6, small / ws * 1e-6, 10 * plane * 16 / pl * 1e-6, pl * 1e3, tl * 1e3);
        CK(hipFree(pb)); CK(hipFree(ps));
    }
    return 0;
}
