// feat_matching.hip -- nn.SpatialMatching on K-plane FEATURE maps (the matcher behind a learned filter stack:
// opticalflow_model.lua:93, tests/time_matching.lua:18 with K = 10, 16x16 window), fast path.
//   out[y][x][dy][dx] = sum_k (in1[k][y][x] - in2[k][y+dy][x+dx])^2,  accumulated over k in order with a separate multiply
// and add -- the CPU restatement's (and the reference-order kernel's) arithmetic, so results are bit-identical to both.
// Mapping as in the raw-frame kernels: lane <-> window cell (64-cell chunks), a thread owns TX = 8 adjacent output columns;
// in1 values are wave-uniform -> scalar loads; the in2 tile (+ window halo) of a slab of KB planes sits in LDS.  A block owns
// TX x TY pixels and deals its TY * ceil(D/64) (row, chunk) tasks to 8 waves, which keep the 8 accumulators of each of their
// tasks in registers across the slabs.  Stores are the tiled kernel's 256-B pieces (same store-pattern bound).
#include "dfe_internal.h"

namespace {

constexpr int FM_TX = 8, FM_TY = 8, FM_NW = 8, FM_MAXT = 8;   // tile, waves, max tasks per wave

typedef const float __attribute__((address_space(4))) *fm_cfptr;
typedef float fm_f8 __attribute__((ext_vector_type(8)));
typedef fm_f8 fm_f8u __attribute__((aligned(4)));

struct FmArgs {
    int K, H1, W1, maxh, maxw, H2, W2, KB, trows, tcols, pitch, nchunks;
};

extern __shared__ __attribute__((aligned(16))) float fm_smem[];

__global__ __launch_bounds__(FM_NW * 64) void feat_matching_kernel(const float *__restrict__ in1, const float *__restrict__ in2,
                                                                  float *__restrict__ out, FmArgs p) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0 = min((int)blockIdx.x * FM_TX, p.W1 - FM_TX), y0 = min((int)blockIdx.y * FM_TY, p.H1 - FM_TY);   // edge tiles shifted inwards
    const int D = p.maxh * p.maxw;
    const int ntasks = FM_TY * p.nchunks;
    float acc[FM_MAXT][FM_TX];
#pragma unroll
    for (int t = 0; t < FM_MAXT; ++t)
#pragma unroll
        for (int x = 0; x < FM_TX; ++x) acc[t][x] = 0.f;
    const long long plane1 = (long long)p.H1 * p.W1, plane2 = (long long)p.H2 * p.W2;

    for (int k0 = 0; k0 < p.K; k0 += p.KB) {
        const int kb = min(p.KB, p.K - k0);
        __syncthreads();                                   // the previous slab is consumed
        // tile rows dealt to the waves, lanes along a row: wave-uniform row arithmetic instead of a div/mod chain per element
        // (that chain cost as many instructions as the matching itself)
        for (int e = threadIdx.x; e < kb * p.trows * p.tcols; e += FM_NW * 64) {
            const int c = e % p.tcols, r = (e / p.tcols) % p.trows, k = e / (p.tcols * p.trows);
            fm_smem[(k * p.trows + r) * p.pitch + c] = in2[(k0 + k) * plane2 + (long long)(y0 + r) * p.W2 + x0 + c];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < FM_MAXT; ++t) {
            const int task = wave + t * FM_NW;
            if (task >= ntasks) break;                     // wave-uniform
            const int row = task / p.nchunks, chunk = task - row * p.nchunks;
            const int d = chunk * 64 + lane;
            const int dc = d < D ? d : D - 1;              // idle lanes shadow the last cell; their stores are masked
            const int dy = dc / p.maxw, dx = dc - dy * p.maxw;
            const float *b0 = fm_smem + (row + dy) * p.pitch + dx;
            // (Tried: in1 through LDS broadcasts instead of scalar loads, two planes per step, operands one plane ahead,
            //  row-wise tile staging with 8 loads in flight -- all within +-10 %: the loop is bound by its K LDS reads
            //  per output, ~0.2 ms at K = 32 / VGA / 16x16 even without bank conflicts.)
            const float *a0 = in1 + (long long)k0 * plane1 + (long long)(y0 + row) * p.W1 + x0;
            for (int k = 0; k < kb; ++k) {
                const fm_f8 a = *(const __attribute__((address_space(4))) fm_f8u *)(fm_cfptr)(a0 + k * plane1);   // s_load_dwordx8
                const float *b = b0 + k * p.trows * p.pitch;
#pragma unroll
                for (int x = 0; x < FM_TX; ++x) {
                    const float df = a[x] - b[x];
                    acc[t][x] = acc[t][x] + df * df;
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < FM_MAXT; ++t) {
        const int task = wave + t * FM_NW;
        if (task >= ntasks) break;
        const int row = task / p.nchunks, chunk = task - row * p.nchunks;
        const int d = chunk * 64 + lane;
        if (d < D) {
            float *o = out + ((long long)(y0 + row) * p.W1 + x0) * D + d;
#pragma unroll
            for (int x = 0; x < FM_TX; ++x) o[(long long)x * D] = acc[t][x];
        }
    }
}

// Row kernel (maxh <= 16 window rows, maxw = MW in {8, 16}): a block is one output row y x 256 columns, WAVE <-> dy, and a
// lane owns PX = 4 adjacent pixels x all MW cells of that window row: PX * MW accumulators in registers across the whole
// k loop.  Per plane the block stages the in2 rows y .. y+maxh-1 (256 + MW - 1 columns) and the in1 row once (double
// buffered, one barrier per plane); a lane then needs 1 + (PX + MW - 1) / 4 ds_read_b128 for PX * MW outputs -- 0.09 LDS
// reads per output and plane instead of 1 (feat_matching_kernel is bound by exactly those reads), and the arithmetic
// (3 VALU per output and plane, the k-ordered separately rounded sum: bit-identical results) becomes the bound.
// A lane's stores are 64-B pieces (one window row of one pixel); the 16 waves of the block write the other rows of the
// same pixels at the same time, so lines complete in L2.
template <int MW>
__global__ __launch_bounds__(1024) void feat_matching_rows_kernel(const float *__restrict__ in1, const float *__restrict__ in2,
                                                                 float *__restrict__ out, FmArgs p) {
#pragma clang fp contract(off)
    constexpr int PX = 4, SEG = 64 * PX, TC = SEG + MW - 1 + 1;    // (+1: row pitch 272 / 264 floats keeps rows 16-B aligned)
    constexpr int NB = (PX + MW - 1 + 3) / 4;                      // b128 reads of the in2 row per lane
    float *tile = fm_smem;                                         // [2][maxh][TC] in2 rows, then [2][SEG] the in1 row
    const int lane = threadIdx.x & 63, dy = threadIdx.x >> 6;      // blockDim.x = 64 * maxh
    const int y = blockIdx.y, x0 = blockIdx.x * SEG;
    const int nthr = 64 * p.maxh;
    float *arow = fm_smem + 2 * p.maxh * TC;
    const long long plane1 = (long long)p.H1 * p.W1, plane2 = (long long)p.H2 * p.W2;
    float acc[PX][MW];
#pragma unroll
    for (int q = 0; q < PX; ++q)
#pragma unroll
        for (int d = 0; d < MW; ++d) acc[q][d] = 0.f;
    // Staging of plane k+1 in two halves around the arithmetic of plane k: its global loads are issued BEFORE (into NS registers per
    // thread: maxh * TC / (64 maxh) <= 5 tile elements + one of the in1 row), the LDS writes come AFTER -- the first version did both
    // in front of the arithmetic, i.e. waited out a memory round trip per plane with nothing to overlap it (16 waves, one block per CU).
    constexpr int NS = (TC + 63) / 64 + 1;
    float sv[NS];
    auto stage_load = [&](int k) {
#pragma unroll
        for (int j = 0; j < NS - 1; ++j) {
            const int e = min((int)threadIdx.x + j * nthr, p.maxh * TC - 1);
            const int r = e / TC, c = e - r * TC;
            sv[j] = in2[k * plane2 + (long long)(y + r) * p.W2 + min(x0 + c, p.W2 - 1)];
        }
        sv[NS - 1] = in1[k * plane1 + (long long)y * p.W1 + min(x0 + min((int)threadIdx.x, SEG - 1), p.W1 - 1)];
    };
    auto stage_commit = [&](int buf) {
        float *t = tile + buf * p.maxh * TC;
#pragma unroll
        for (int j = 0; j < NS - 1; ++j) {
            const int e = threadIdx.x + j * nthr;
            if (e < p.maxh * TC) t[e] = sv[j];
        }
        if ((int)threadIdx.x < SEG) arow[buf * SEG + threadIdx.x] = sv[NS - 1];
    };
    stage_load(0);
    stage_commit(0);
    for (int k = 0; k < p.K; ++k) {
        __syncthreads();                                           // plane k is staged; plane k-1's buffer is free
        if (k + 1 < p.K) stage_load(k + 1);
        const float4 *br = reinterpret_cast<const float4 *>(tile + ((k & 1) * p.maxh + dy) * TC + PX * lane);
        const float4 a4 = *reinterpret_cast<const float4 *>(arow + (k & 1) * SEG + PX * lane);
        float b[4 * NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float4 v = br[j];
            b[4 * j] = v.x; b[4 * j + 1] = v.y; b[4 * j + 2] = v.z; b[4 * j + 3] = v.w;
        }
        const float a[PX] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int q = 0; q < PX; ++q)
#pragma unroll
            for (int d = 0; d < MW; ++d) {
                const float df = a[q] - b[q + d];
                acc[q][d] = acc[q][d] + df * df;
            }
        if (k + 1 < p.K) stage_commit((k + 1) & 1);
    }
    // Copy-out through LDS: a pixel's window (maxh x MW cells) is one contiguous run of the output, but its rows sit in maxh
    // different waves.  For each of the lane's PX pixels in turn, every wave deposits its row into [lane][dy][MW]; after a
    // barrier the waves copy whole windows out, one 16-B piece per lane and window: each store instruction writes contiguous
    // memory instead of 64 scattered 16-B pieces (the first version of this kernel was bound by exactly those: 0.45 ms at
    // K = 32 / VGA against 0.9 us of arithmetic per plane).
    float *win = arow + 2 * SEG;                                   // [64][maxh * MW]
    const int WN = p.maxh * MW;                                    // floats per window
#pragma unroll
    for (int q = 0; q < PX; ++q) {
        __syncthreads();                                           // (q = 0: the last plane's tile reads; q > 0: the previous copy)
        float4 *wd = reinterpret_cast<float4 *>(win + lane * WN + dy * MW);
#pragma unroll
        for (int j = 0; j < MW / 4; ++j) wd[j] = make_float4(acc[q][4 * j], acc[q][4 * j + 1], acc[q][4 * j + 2], acc[q][4 * j + 3]);
        __syncthreads();
        const int n4 = WN / 4;                                     // 16-B pieces per window
        for (int pxl = dy; pxl < 64; pxl += p.maxh) {              // wave dy copies the windows of lanes dy, dy + maxh, ...
            const int x = x0 + PX * pxl + q;
            if (x >= p.W1) break;                                  // (wave-uniform)
            const float4 *src = reinterpret_cast<const float4 *>(win + pxl * WN);
            float4 *dst = reinterpret_cast<float4 *>(out + ((long long)y * p.W1 + x) * WN);
            for (int i = lane; i < n4; i += 64) dst[i] = src[i];
        }
    }
}

// ---- windows of exactly one chunk (maxh * maxw == 64: the pyramid's 8 x 8), K <= 16 planes, several independent pairs per launch ----
// The matcher behind the learned filters of getModelMultiscale (every scale's nn.SpatialMatching(maxh, maxw) on K-plane features).
// feat_matching_kernel above spends its time waiting: one s_load_dwordx8 per plane in front of the arithmetic that needs it (an
// SMEM round trip per plane, K of them in a row), 256-B dword stores.  Here: lane <-> cell, a thread owns 8 adjacent columns, a block
// of 8 waves a tile of 8 columns x 16 rows (two rows per wave, 16 accumulators); ALL K planes of the in2 tile sit in LDS (pitch ==
// maxw mod 32: the 64 cells of a wave hit 64 different banks in two passes); the in1 scalars of planes k+2, k+3 are requested
// behind the LDS reads of planes k, k+1 and in front of their arithmetic, so the SMEM round trip is covered by 48 VALU
// instructions and never in flight while LDS data is waited for (SMEM returns out of order: any LDS wait would have to drain
// it); a task row (8 pixels x 256 B) leaves through a 2-KB transpose of the wave as two 1-KB stores -- or, for fp16 volumes, is
// converted (cost * scale, round to nearest even) and leaves as one.  Same k-ordered, separately rounded sums: bit-identical
// to the other matchers and to the CPU loop.
struct FmBatch {
    const float *in1[DFE_MAX_RATIOS], *in2[DFE_MAX_RATIOS];
    float *out[DFE_MAX_RATIOS];
    int H1[DFE_MAX_RATIOS], W1[DFE_MAX_RATIOS];
};
constexpr int F64_TY = 16;
// MODE 0: the volume is stored.  MODE 1 / 2 (8 x 8 windows, one pair): the task rows -- 8 pixels x 64 cells, lane <-> cell, exactly the
// tiled SSD kernel's -- go through fine_epilogue (dfe_internal.h) instead: the finest pyramid scale (1) or a scale between (2) without
// its volume, as in ssd_cv_tiled_fine_kernel.  F16: the costs are rounded as the stored fp16 volume would hold them.
template <int MODE, bool F16>
__device__ __forceinline__ void feat_matching_win64_body(const FmBatch &fb, int K, int maxh, int maxw, int pitch, float f16_scale, int nt, const CvFineArgs *fine) {
#pragma clang fp contract(off)
    const int z = gridDim.z - 1 - blockIdx.z;              // (the smallest pair first: its few blocks must not form the tail)
    const int H1 = fb.H1[z], W1 = fb.W1[z];
    if ((int)blockIdx.x * FM_TX >= W1 || (int)blockIdx.y * F64_TY >= H1) return;   // block-uniform
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0 = min((int)blockIdx.x * FM_TX, W1 - FM_TX), y0 = min((int)blockIdx.y * F64_TY, H1 - F64_TY);   // edge tiles shifted inwards
    const int W2 = W1 + maxw - 1, H2 = H1 + maxh - 1;
    const int trows = F64_TY + maxh - 1, tcols = FM_TX + maxw - 1;
    const long long plane1 = (long long)H1 * W1, plane2 = (long long)H2 * W2;
    const float *__restrict__ in2 = fb.in2[z];
    {
        // staging: a load instruction covers FOUR tile rows (16 lanes each: tcols <= 16 here), and a thread issues all of its loads
        // before the first LDS write -- one exposed memory latency per block instead of one per tile row (the first version, a row
        // per iteration with the LDS write behind each load, spent 80 % of the launch waiting here)
        constexpr int NB = 8;                                  // loads in flight per thread
        const int rsub = threadIdx.x >> 4, c = threadIdx.x & 15;     // 32 row slots x 16 columns per pass
        const int nrows = K * trows;
        for (int r0 = 0; r0 < nrows; r0 += 32 * NB) {
            float v[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int r = min(r0 + i * 32 + rsub, nrows - 1);
                const int k = r / trows, rr = r - k * trows;
                v[i] = in2[k * plane2 + (long long)(y0 + rr) * W2 + x0 + min(c, tcols - 1)];
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int r = r0 + i * 32 + rsub;
                if (r < nrows && c < tcols) fm_smem[r * pitch + c] = v[i];
            }
        }
    }
    __syncthreads();
    const int dy = lane / maxw, dx = lane - dy * maxw;
    const int pl = trows * pitch;                          // floats per plane of the tile
    float acc[2][FM_TX];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int x = 0; x < FM_TX; ++x) acc[t][x] = 0.f;
    const float *a_base = fb.in1[z] + (long long)y0 * W1 + x0;
    // (a row base per task + a 32-bit plane offset, which the scalar load takes as its offset operand: the 64-bit k * plane + row * W
    //  of the first version was a dozen scalar instructions per load; the launcher checks K * plane bytes < 2^32)
    const char *const rb0 = reinterpret_cast<const char *>(a_base + (long long)wave * W1);
    const char *const rb1 = reinterpret_cast<const char *>(a_base + (long long)(wave + 8) * W1);
    const unsigned p1b = (unsigned)plane1 * 4u;
    auto lda = [&](int k, int row) -> fm_f8 {
        return *(const __attribute__((address_space(4))) fm_f8u *)(fm_cfptr)((row == wave ? rb0 : rb1) + (unsigned long long)((unsigned)k * p1b));   // s_load_dwordx8
    };
    // Two planes per step, the frame-1 values of the NEXT two planes requested (scalar loads) behind this step's LDS reads.  The steps
    // alternate between two sets of scalars (a / an) instead of copying the new set over the old one, and the loop runs over whole pairs
    // of planes only, an odd last plane behind it: the first version's loop body was 96 arithmetic instructions in 300 -- 32 scalar moves
    // for the copy, a select and a branch per element for `is there a second plane` (profiles/r04_au: 1.28e7 scalar instructions per
    // launch for 2.33e7 vector ones), and these kernels pay for every instruction issued.
    fm_f8 a[2][2], an[2][2];                               // [task][plane parity]
#pragma unroll
    for (int t = 0; t < 2; ++t) { a[t][0] = lda(0, wave + 8 * t); a[t][1] = lda(K > 1 ? 1 : 0, wave + 8 * t); }
    auto step2 = [&](const fm_f8 (&cur)[2][2], fm_f8 (&nxt)[2][2], int k) __attribute__((always_inline)) {
        float b[2][2][FM_TX];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float *bp = fm_smem + (wave + 8 * t + dy) * pitch + dx + k * pl;
#pragma unroll
            for (int x = 0; x < FM_TX; ++x) { b[t][0][x] = bp[x]; b[t][1][x] = bp[pl + x]; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this step's LDS data (and the scalars requested a step ago) are in
        __builtin_amdgcn_sched_barrier(0);
        const int k2 = min(k + 2, K - 1), k3 = min(k + 3, K - 1);
#pragma unroll
        for (int t = 0; t < 2; ++t) { nxt[t][0] = lda(k2, wave + 8 * t); nxt[t][1] = lda(k3, wave + 8 * t); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int x = 0; x < FM_TX; ++x) {
                const float df = cur[t][0][x] - b[t][0][x];
                acc[t][x] = acc[t][x] + df * df;
            }
#pragma unroll
            for (int x = 0; x < FM_TX; ++x) {
                const float df = cur[t][1][x] - b[t][1][x];
                acc[t][x] = acc[t][x] + df * df;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    int k = 0;
    for (; k + 3 < K; k += 4) { step2(a, an, k); step2(an, a, k + 2); }
    if (k + 1 < K) {
        step2(a, an, k);
        k += 2;
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t][0] = an[t][0];
    }
    if (k < K) {                                           // an odd last plane: its frame-1 values are a[.][0] (requested as min(k, K - 1) above)
        float b[2][FM_TX];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float *bp = fm_smem + (wave + 8 * t + dy) * pitch + dx + k * pl;
#pragma unroll
            for (int x = 0; x < FM_TX; ++x) b[t][x] = bp[x];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int x = 0; x < FM_TX; ++x) {
                const float df = a[t][0][x] - b[t][x];
                acc[t][x] = acc[t][x] + df * df;
            }
    }
    if constexpr (MODE != 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) fine_epilogue<FM_TX, F16, MODE == 2>(acc[t], lane, y0 + wave + 8 * t, x0, W1, *fine);
        return;
    }
    // copy-out through the wave's own 2-KB scratch (the tile is no longer read by THIS wave; other waves' rows are elsewhere)
    __syncthreads();                                       // every wave is done with the tile: its space is the scratch now
    float *xp = fm_smem + wave * (FM_TX * 64);
    typedef float f4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int y = y0 + wave + 8 * t;
#pragma unroll
        for (int x = 0; x < FM_TX; ++x) xp[x * 64 + lane] = acc[t][x];
        const long long pix = (long long)y * W1 + x0;
        if (f16_scale != 0.f) {
            const f4v lo = *reinterpret_cast<const f4v *>(xp + 8 * lane), hi = *reinterpret_cast<const f4v *>(xp + 8 * lane + 4);
            typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
            union { h2_t h[4]; f4v v; } u;
            u.h[0] = h2_t{(_Float16)(lo[0] * f16_scale), (_Float16)(lo[1] * f16_scale)};
            u.h[1] = h2_t{(_Float16)(lo[2] * f16_scale), (_Float16)(lo[3] * f16_scale)};
            u.h[2] = h2_t{(_Float16)(hi[0] * f16_scale), (_Float16)(hi[1] * f16_scale)};
            u.h[3] = h2_t{(_Float16)(hi[2] * f16_scale), (_Float16)(hi[3] * f16_scale)};
            const char *orow = (const char *)fb.out[z] + pix * 128;
            if (nt) asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"((unsigned)lane * 16u), "v"(u.v), "s"(orow) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"((unsigned)lane * 16u), "v"(u.v), "s"(orow) : "memory");
        } else {
            const f4v lo = *reinterpret_cast<const f4v *>(xp + 4 * lane), hi = *reinterpret_cast<const f4v *>(xp + 256 + 4 * lane);
            const char *orow = (const char *)(fb.out[z] + pix * 64);
            if (nt) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\tglobal_store_dwordx4 %0, %3, %2 offset:1024 nt" ::"v"((unsigned)lane * 16u), "v"(lo), "s"(orow), "v"(hi) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, %2\n\tglobal_store_dwordx4 %0, %3, %2 offset:1024" ::"v"((unsigned)lane * 16u), "v"(lo), "s"(orow), "v"(hi) : "memory");
        }
    }
}
__global__ __launch_bounds__(512) void feat_matching_win64_kernel(FmBatch fb, int K, int maxh, int maxw, int pitch, float f16_scale, int nt) {
    feat_matching_win64_body<0, false>(fb, K, maxh, maxw, pitch, f16_scale, nt, nullptr);
}
template <int MODE, bool F16>
__global__ __launch_bounds__(512) void feat_matching_win64_fine_kernel(FmBatch fb, int K, int pitch, CvFineArgs fine) {
    feat_matching_win64_body<MODE, F16>(fb, K, 8, 8, pitch, 0.f, 0, &fine);
}

}  // namespace

// The ctx- and window-level conditions of dfe_feat_matching_win64_batch (everything but the per-pair sizes): the multiscale launcher
// plans its fused scales with the SAME predicate the launcher applies, so a plan never meets a refusal (round-3 advisor: with
// dfe_set_cost_volume_kernel(1) or fm64 = 0 the fused second scale was planned and then refused).
bool dfe_feat_matching_win64_ok(const dfe_ctx *ctx, int K, int maxh, int maxw) {
    if (maxh * maxw != 64 || K < 1 || K > 16 || ctx->cv_mode == 1 || ctx->opt[DFE_OPT_FM64] == 0) return false;
    const int tcols = FM_TX + maxw - 1, trows = F64_TY + maxh - 1;
    if (tcols > 16) return false;                          // (the staging deals 16 columns per tile row)
    int pitch = tcols;
    while ((pitch - maxw) % 32 != 0) ++pitch;
    return (size_t)K * trows * pitch * sizeof(float) <= 64 * 1024;
}

// n pairs (pyramid scales) of K-plane features through one launch of the one-chunk matcher; out[i] [H1][W1][64] f32, or half volumes
// (half(cost * f16_scale)) when f16_scale != 0.  *handled = false: not this kernel's shape (the caller launches pair by pair).
// fine != NULL (n == 1, 8 x 8 windows): no volume -- the pair's task rows go through the fused pyramid epilogue (fine->casc == NULL: the
// finest scale; else a scale between), fine->f16_scale as in cv_frames_finest_fused; out is not used.
int dfe_feat_matching_win64_batch(dfe_ctx *ctx, int n, const float *const *in1, const float *const *in2, int K, const int *H1, const int *W1, int maxh,
                                  int maxw, float *const *out, float f16_scale, bool *handled, const CvFineArgs *fine) {
    *handled = false;
    if (!dfe_feat_matching_win64_ok(ctx, K, maxh, maxw) || n < 1 || n > DFE_MAX_RATIOS) return DFE_OK;
    if (fine && (n != 1 || maxh != 8 || maxw != 8 || (fine->pcasc && ((H1[0] | W1[0]) & 1)))) return DFE_OK;
    FmBatch fb;
    int gx = 0, gy = 0;
    size_t vol = 0;
    for (int i = 0; i < n; ++i) {
        if (H1[i] < F64_TY || W1[i] < FM_TX || ((uintptr_t)in1[i] & 3) || (!fine && ((uintptr_t)out[i] & 15))) return DFE_OK;
        if ((long long)K * H1[i] * W1[i] * 4 >= (1ll << 32)) return DFE_OK;           // (32-bit plane offsets of the frame-1 scalar loads)
        fb.in1[i] = in1[i]; fb.in2[i] = in2[i]; fb.out[i] = fine ? nullptr : out[i]; fb.H1[i] = H1[i]; fb.W1[i] = W1[i];
        gx = gx > dfe_cdiv(W1[i], FM_TX) ? gx : dfe_cdiv(W1[i], FM_TX);
        gy = gy > dfe_cdiv(H1[i], F64_TY) ? gy : dfe_cdiv(H1[i], F64_TY);
        vol += (size_t)H1[i] * W1[i] * 64 * (f16_scale != 0.f ? 2 : 4);
    }
    const int tcols = FM_TX + maxw - 1, trows = F64_TY + maxh - 1;
    if (tcols > 16) return DFE_OK;                         // (the staging deals 16 columns per tile row)
    int pitch = tcols;
    while ((pitch - maxw) % 32 != 0) ++pitch;              // pitch == maxw (mod 32): conflict-free for the lane <-> (dy, dx) reads
    size_t lds = (size_t)K * trows * pitch * sizeof(float);
    if (lds < (size_t)8 * FM_TX * 64 * sizeof(float)) lds = (size_t)8 * FM_TX * 64 * sizeof(float);   // the copy-out scratch reuses the tile
    if (lds > 64 * 1024) return DFE_OK;
    if (fine) {
        const bool h16 = fine->f16_scale != 0.f, mid = fine->casc != nullptr;
        auto kern = mid ? (h16 ? feat_matching_win64_fine_kernel<2, true> : feat_matching_win64_fine_kernel<2, false>)
                        : (h16 ? feat_matching_win64_fine_kernel<1, true> : feat_matching_win64_fine_kernel<1, false>);
        DFE_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        {
            DfeProfScope prof(ctx);
            hipLaunchKernelGGL(kern, dim3(gx, gy, 1), dim3(512), lds, ctx->stream, fb, K, pitch, *fine);
        }
        DFE_LAUNCH_CHECK(ctx);
        ctx->last_kernel = mid ? (h16 ? "feat_matching_win64_mid_kernel_f16" : "feat_matching_win64_mid_kernel") : h16 ? "feat_matching_win64_fine_kernel_f16" : "feat_matching_win64_fine_kernel";
        *handled = true;
        return DFE_OK;
    }
    DFE_HIP(ctx, hipFuncSetAttribute((const void *)feat_matching_win64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nt = vol > ((size_t)160 << 20);              // (volumes that do not stay in the memory-side cache: non-temporal stores)
    {
        DfeProfScope prof(ctx);
        hipLaunchKernelGGL(feat_matching_win64_kernel, dim3(gx, gy, n), dim3(512), lds, ctx->stream, fb, K, maxh, maxw, pitch, f16_scale, nt);
    }
    DFE_LAUNCH_CHECK(ctx);
    ctx->last_kernel = f16_scale != 0.f ? "feat_matching_win64_kernel_f16" : "feat_matching_win64_kernel";
    *handled = true;
    return DFE_OK;
}

// *handled stays false when the shape has no fast instantiation (the caller falls back to the reference-order kernel)
int dfe_feat_matching_fast(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh, int maxw, float *out,
                           bool *handled) {
    *handled = false;
    const int D = maxh * maxw, nchunks = (D + 63) / 64;
    if (ctx->cv_mode == 1) return DFE_OK;
    if (maxh * maxw == 64 && K <= 16 && ctx->cv_mode != 2) {   // one-chunk windows (the pyramid's 8 x 8): the prefetching, transposing matcher
        int rc = dfe_feat_matching_win64_batch(ctx, 1, &in1, &in2, K, &H1, &W1, maxh, maxw, &out, 0.f, handled);
        if (rc != DFE_OK || *handled) return rc;
    }
    {   // 16- / 17-wide windows on frames at least 253 pixels wide: the flat-tile kernel (feat_matching_flat.hip)
        int rc = dfe_feat_matching_flat(ctx, in1, in2, K, H1, W1, maxh, maxw, out, handled);
        if (rc != DFE_OK || *handled) return rc;
    }
    // The row kernel pays a barrier and a tile refill per plane and only fills its 256-column blocks on wide frames: measured
    // 625 x 465, 16 x 16: K = 32 0.37 ms against 0.55 ms for the chunk kernel below, K = 10 0.168 against 0.194 (with the next plane's
    // loads in flight behind the arithmetic; 0.46 / 0.23 before); K = 10, 293 x 153: 0.064 against 0.039 ms.
    // (W1 == 1: the patch-mode call of the trainers, any K -- the chunk kernel needs 8 x 8 pixels.)
    bool rows_pays = (K >= 8 && W1 >= 400) || W1 < FM_TX || H1 < FM_TY;
    rows_pays = ctx->opt_bool(DFE_OPT_FM_ROWS, rows_pays);   // tuning
    if (rows_pays && (maxw == 16 || maxw == 8) && maxh >= 4 && maxh <= 16 && ctx->cv_mode != 2 && ((uintptr_t)out & 15) == 0) {
        FmArgs a{};
        a.K = K; a.H1 = H1; a.W1 = W1; a.maxh = maxh; a.maxw = maxw; a.H2 = H1 + maxh - 1; a.W2 = W1 + maxw - 1;
        const int TC = 256 + maxw;
        const size_t lds = ((size_t)2 * maxh * TC + 2 * 256 + (size_t)64 * maxh * maxw) * sizeof(float);
        dim3 grid(dfe_cdiv(W1, 256), H1);
        auto kern = maxw == 16 ? feat_matching_rows_kernel<16> : feat_matching_rows_kernel<8>;
        DFE_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        {
            DfeProfScope prof(ctx);
            hipLaunchKernelGGL(kern, grid, dim3(64 * maxh), lds, ctx->stream, in1, in2, out, a);
        }
        DFE_LAUNCH_CHECK(ctx);
        ctx->last_kernel = "feat_matching_rows_kernel";
        *handled = true;
        return DFE_OK;
    }
    if (W1 < FM_TX || H1 < FM_TY || FM_TY * nchunks > FM_MAXT * FM_NW) return DFE_OK;
    if (((uintptr_t)in1 & 3) != 0) return DFE_OK;
    FmArgs a;
    a.K = K; a.H1 = H1; a.W1 = W1; a.maxh = maxh; a.maxw = maxw; a.H2 = H1 + maxh - 1; a.W2 = W1 + maxw - 1;
    a.trows = FM_TY + maxh - 1; a.tcols = FM_TX + maxw - 1;
    a.pitch = a.tcols | 1;                                  // odd pitch: the lanes behind a dy-row jump land on other banks
    a.nchunks = nchunks;
    const size_t per_plane = (size_t)a.trows * a.pitch * sizeof(float);
    int kb = (int)((48 * 1024) / per_plane);
    if (kb < 1) return DFE_OK;
    a.KB = kb < K ? kb : K;
    const size_t lds = (size_t)a.KB * per_plane;
    DFE_HIP(ctx, hipFuncSetAttribute((const void *)feat_matching_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(dfe_cdiv(W1, FM_TX), dfe_cdiv(H1, FM_TY));
    {
        DfeProfScope prof(ctx);
        hipLaunchKernelGGL(feat_matching_kernel, grid, dim3(FM_NW * 64), lds, ctx->stream, in1, in2, out, a);
    }
    DFE_LAUNCH_CHECK(ctx);
    ctx->last_kernel = "feat_matching_kernel";
    *handled = true;
    return DFE_OK;
}
