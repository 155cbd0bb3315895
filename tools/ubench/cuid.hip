// Which blocks share a CU?  512 blocks of 512 threads and 69 KB of LDS each (the half-tile launch of the flat matcher): every block
// records its XCC_ID and HW_ID.  hipcc --offload-arch=gfx950 -O3 -o cuid cuid.hip && ./cuid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
extern __shared__ float smem[];
__global__ void who(unsigned *out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    smem[threadIdx.x] = 1.f;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(127);   // keep every block resident until all have started
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main() {
    const int nb = 512;
    unsigned *d; hipMalloc(&d, nb * 8);
    hipFuncSetAttribute((const void *)who, hipFuncAttributeMaxDynamicSharedMemorySize, 69 * 1024);
    hipLaunchKernelGGL(who, dim3(nb), dim3(512), 69 * 1024, 0, d, 40);
    std::vector<unsigned> h(2 * nb);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < nb; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const unsigned cuid = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[(xcc << 12) | (se << 8) | (sh << 4) | cuid].push_back(b);
    }
    printf("%zu distinct CUs\n", cu.size());
    int n = 0;
    for (auto &kv : cu) {
        if (n++ < 24 || kv.second.size() != 2) { printf("xcc %u se %u sh %u cu %2u:", kv.first >> 12, (kv.first >> 8) & 7, (kv.first >> 4) & 1, kv.first & 15); for (int b : kv.second) printf(" %d", b); printf("\n"); }
    }
    return 0;
}
