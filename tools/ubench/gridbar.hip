// What does a grid-wide phase boundary cost INSIDE one launch, against a kernel boundary on one stream?  ("VGA {1,2,4}: one
// cooperative launch with grid-wide phases ... to drop three launch boundaries".)  Both move 64 KB per block from a producer phase to a
// consumer phase through global memory, as the pyramid's phases do (volumes -> cascade):
//   A  N+1 kernels on one stream, phase p reads what phase p-1 wrote (another block's data);
//   B  one launch of 256 resident blocks, N grid barriers: every wave's stores drained (vmcnt(0)), __syncthreads, agent-scope release by
//      one lane, an atomic counter all blocks poll, agent-scope acquire, __syncthreads (MI355X_MICROARCH.md, cross-workgroup hand-off).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int NB = 256, NT = 256, PER = 64 * 1024 / 4 / NT;   // 64 KB per block and phase
__device__ __forceinline__ void phase(float *buf, int p, int blk) {
    const float *src = buf + (size_t)((p & 1) ^ 1) * NB * NT * PER + (size_t)((blk + 37) % NB) * NT * PER;   // another block's output of the previous phase
    float *dst = buf + (size_t)(p & 1) * NB * NT * PER + (size_t)blk * NT * PER;
    for (int i = 0; i < PER; ++i) dst[i * NT + threadIdx.x] = src[i * NT + threadIdx.x] + 1.0f;
}
__global__ __launch_bounds__(NT) void k_phase(float *buf, int p) { phase(buf, p, blockIdx.x); }
__global__ __launch_bounds__(NT) void k_coop(float *buf, unsigned *counter, int nph) {
    for (int p = 0; p < nph; ++p) {
        phase(buf, p, blockIdx.x);
        if (p + 1 == nph) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(p + 1) * gridDim.x;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
}
int main() {
    float *buf; unsigned *cnt;
    hipMalloc(&buf, (size_t)2 * NB * NT * PER * 4); hipMemset(buf, 0, (size_t)2 * NB * NT * PER * 4);
    hipMalloc(&cnt, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int NPH = 4, REP = 200;
    for (int variant = 0; variant < 2; ++variant) {
        float best = 1e9;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0);
            for (int it = 0; it < REP; ++it) {
                if (variant == 0) { for (int p = 0; p < NPH; ++p) hipLaunchKernelGGL(k_phase, dim3(NB), dim3(NT), 0, 0, buf, p); }
                else { hipMemsetAsync(cnt, 0, 4, 0); hipLaunchKernelGGL(k_coop, dim3(NB), dim3(NT), 0, 0, buf, cnt, NPH); }
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%s: %d phases of 64 KB per block, 256 blocks: %.2f us per %d-phase step\n", variant == 0 ? "A  one kernel per phase          " : "B  one launch, grid-wide barriers", NPH, best * 1e3 / REP, NPH);
    }
    // phases alone (no dependency): one kernel doing all phases without barriers (wrong results, timing only)
    {
        float best = 1e9;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0);
            for (int it = 0; it < REP; ++it) hipLaunchKernelGGL(k_phase, dim3(NB), dim3(NT), 0, 0, buf, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("   one phase kernel back to back: %.2f us per launch\n", best * 1e3 / REP);
    }
    return 0;
}
