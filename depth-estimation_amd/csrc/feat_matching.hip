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
    auto stage = [&](int k, int buf) {
        float *t = tile + buf * p.maxh * TC;
        for (int e = threadIdx.x; e < p.maxh * TC; e += nthr) {
            const int r = e / TC, c = e - r * TC;
            const int x = min(x0 + c, p.W2 - 1);
            t[e] = in2[k * plane2 + (long long)(y + r) * p.W2 + x];
        }
        for (int e = threadIdx.x; e < SEG; e += nthr) arow[buf * SEG + e] = in1[k * plane1 + (long long)y * p.W1 + min(x0 + e, p.W1 - 1)];
    };
    stage(0, 0);
    for (int k = 0; k < p.K; ++k) {
        __syncthreads();                                           // plane k is staged; plane k-1's buffer is free
        if (k + 1 < p.K) stage(k + 1, (k + 1) & 1);
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

}  // namespace

// *handled stays false when the shape has no fast instantiation (the caller falls back to the reference-order kernel)
int dfe_feat_matching_fast(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh, int maxw, float *out,
                           bool *handled) {
    *handled = false;
    const int D = maxh * maxw, nchunks = (D + 63) / 64;
    if (ctx->cv_mode == 1) return DFE_OK;
    // The row kernel pays a barrier and a tile refill per plane and only fills its 256-column blocks on wide frames: measured
    // K = 32, 625 x 465, 16 x 16: 0.46 ms against 0.54 ms for the chunk kernel below; K = 10, 293 x 153: 0.064 against 0.039 ms.
    // (W1 == 1: the patch-mode call of the trainers, any K -- the chunk kernel needs 8 x 8 pixels.)
    const bool rows_pays = (K >= 24 && W1 >= 400) || W1 < FM_TX || H1 < FM_TY;
    if (rows_pays && (maxw == 16 || maxw == 8) && maxh >= 2 && maxh <= 16 && ctx->cv_mode != 2 && ((uintptr_t)out & 15) == 0) {
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
