// filters.hip -- A15 / next-row N1: the learned patch-feature stack in front of the matcher (getFilter,
// opticalflow_model.lua:45-79, radial/radial_opticalflow_network.lua:6-30): nn.SpatialConvolution, nn.SpatialConvolutionMap,
// nn.Tanh.  Direct form, one thread per output element, the accumulation order of the CPU restatement (input plane or
// connection, then ky, kx) so results are bit-identical to it.  Untuned: this is the one place on the path where an
// implicit-GEMM MFMA kernel applies; the layers are a few percent of a matcher pass at the reference's sizes.
#include "dfe_internal.h"
#include <cstdlib>

namespace {

__global__ void conv_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias, int nIn, int nOut,
                            int H, int W, int kH, int kW, float *__restrict__ out) {
#pragma clang fp contract(off)
    const int Ho = H - kH + 1, Wo = W - kW + 1;
    const long long n = (long long)nOut * Ho * Wo;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(e % Wo);
        const long long t = e / Wo;
        const int y = (int)(t % Ho), o = (int)(t / Ho);
        float s = bias ? bias[o] : 0.f;
        for (int i = 0; i < nIn; ++i)
            for (int u = 0; u < kH; ++u)
                for (int v = 0; v < kW; ++v)
                    s = s + w[(((long long)o * nIn + i) * kH + u) * kW + v] * in[((long long)i * H + y + u) * W + x + v];
        out[e] = s;
    }
}

__global__ void conv_map_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
                                const int *__restrict__ conn, int nConn, int nOut, int H, int W, int kH, int kW, float *__restrict__ out) {
#pragma clang fp contract(off)
    const int Ho = H - kH + 1, Wo = W - kW + 1;
    const long long n = (long long)nOut * Ho * Wo;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(e % Wo);
        const long long t = e / Wo;
        const int y = (int)(t % Ho), o = (int)(t / Ho);
        float s = bias ? bias[o] : 0.f;
        for (int c = 0; c < nConn; ++c) {
            if (conn[2 * c + 1] - 1 != o) continue;
            const int i = conn[2 * c] - 1;
            for (int u = 0; u < kH; ++u)
                for (int v = 0; v < kW; ++v) s = s + w[((long long)c * kH + u) * kW + v] * in[((long long)i * H + y + u) * W + x + v];
        }
        out[e] = s;
    }
}

// the same two kernels with nn.Tanh applied to the result (one launch per layer of a filter stack inside the one-call pipelines)
template <bool TANH>
__global__ void conv_layer_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias, const int *__restrict__ conn,
                                  int nConn, int nIn, int nOut, int H, int W, int kH, int kW, float *__restrict__ out) {
#pragma clang fp contract(off)
    const int Ho = H - kH + 1, Wo = W - kW + 1;
    const long long n = (long long)nOut * Ho * Wo;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(e % Wo);
        const long long t = e / Wo;
        const int y = (int)(t % Ho), o = (int)(t / Ho);
        float s = bias ? bias[o] : 0.f;
        if (conn) {
            for (int c = 0; c < nConn; ++c) {
                if (conn[2 * c + 1] - 1 != o) continue;
                const int i = conn[2 * c] - 1;
                for (int u = 0; u < kH; ++u)
                    for (int v = 0; v < kW; ++v) s = s + w[((long long)c * kH + u) * kW + v] * in[((long long)i * H + y + u) * W + x + v];
            }
        } else {
            for (int i = 0; i < nIn; ++i)
                for (int u = 0; u < kH; ++u)
                    for (int v = 0; v < kW; ++v)
                        s = s + w[(((long long)o * nIn + i) * kH + u) * kW + v] * in[((long long)i * H + y + u) * W + x + v];
        }
        out[e] = TANH ? tanhf(s) : s;
    }
}

__global__ void tanh_kernel(const float *__restrict__ in, long long n, float *__restrict__ out) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) out[e] = tanhf(in[e]);
}

int grid_n(long long n) {
    long long b = (n + 255) / 256;
    if (b > 256 * 32) b = 256 * 32;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

// ---- the same layer for a BATCH of independent inputs in one launch (both frames of every pyramid scale), LDS-tiled ----------------
// A block = 32 x 8 threads owns 128 x 8 output pixels of one entry; a thread owns a strip of 4 adjacent pixels and a group of NT
// output planes (blockIdx.y = entry * groups + group).  The block stages the (8 + kH - 1) x (128 + KW - 1) tile of every input
// plane in LDS once; per (input plane, kernel row) a thread reads its 4 + KW - 1 tile values (two ds_read_b128) and the KW weights
// of each of its NT planes through scalar loads (the weight index is wave-uniform), i.e. 4 KW NT multiply-adds per 2 LDS reads.
// Accumulation order per output: bias, then (input plane, ky, kx) with separately rounded multiply and add -- exactly
// conv_kernel's, so the results are bit-identical to it and to the CPU loop.
struct ConvBatch {
    const float *in[2 * DFE_MAX_RATIOS];
    float *out[2 * DFE_MAX_RATIOS];
    const float *w[2 * DFE_MAX_RATIOS], *bias[2 * DFE_MAX_RATIOS];
    int H[2 * DFE_MAX_RATIOS], W[2 * DFE_MAX_RATIOS];
    unsigned pitch[2 * DFE_MAX_RATIOS], plane[2 * DFE_MAX_RATIOS];   // floats between rows / planes of in[e] (W, H * W unless the input is a view)
    int blk0[2 * DFE_MAX_RATIOS + 1];   // first block of entry e (tiles x output groups each): the grid holds no idle blocks
    int n;
};
constexpr int CB_PX = 4;   // (tile: 128 x 8 or 64 x 16 outputs, 256 threads)
extern __shared__ __attribute__((aligned(16))) float conv_smem[];

// NARROW: tiles of 64 x 16 outputs (16 x 16 threads) instead of 128 x 8: less halo per output for large kernels (the launcher's
// rule and the measurement behind it: launch_conv_batch).  Same strips, same order of operations per output: the results do not
// depend on the shape.
// The LDS-DMA requests of one tile row of PITCH floats as ONE assembly block: lane l fetches the floats at sbase + voff{0,1,2} (bytes)
// into LDS at lds_dst + 4 l, + 256 + 4 l, + 512 + 4 l; the last request covers PITCH - 64 (N - 1) lanes under an execution mask set
// and restored by scalar instructions; M0 (the destination base) saved and restored once (cdna_hip_programming.md, LDS-DMA recipe).
// Called with all 64 lanes active.  (One block per request, a branch around the partial one: twice the scalar instructions -- and in
// these kernels every instruction costs issue time)
template <int PITCH> __device__ __forceinline__ void cb_glds_row(unsigned v0, unsigned v1, unsigned v2, const void *sbase, const float *lds_dst) {
    const unsigned la = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)lds_dst;
    unsigned keep;
    unsigned long long ex;
    static_assert(PITCH > 64 && PITCH <= 192, "two or three requests per row");
    if constexpr (PITCH > 128) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dword %2, %6\n\t"
                     "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %3, %6\n\t"
                     "s_add_u32 m0, m0, 0x100\n\ts_and_saveexec_b64 %1, %7\n\tglobal_load_lds_dword %4, %6\n\t"
                     "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(ex) : "v"(v0), "v"(v1), "v"(v2), "s"(la), "s"(sbase), "s"((1ull << (PITCH - 128)) - 1) : "memory", "scc");
    } else {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, %5\n\t"
                     "s_add_u32 m0, m0, 0x100\n\ts_and_saveexec_b64 %1, %6\n\tglobal_load_lds_dword %3, %5\n\t"
                     "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(ex) : "v"(v0), "v"(v1), "s"(la), "s"(sbase), "s"(PITCH == 128 ? ~0ull : (1ull << (PITCH - 64)) - 1) : "memory", "scc");
    }
}

template <int KW, int NT, bool TANH, bool NARROW = false>
__global__ __launch_bounds__(256) void conv_batch_kernel(ConvBatch cb, int nIn, int nOut, int kH, int groups) {
#pragma clang fp contract(off)
    constexpr int CB_TW = NARROW ? 64 : 128, CB_TH = NARROW ? 16 : 8, TXN = CB_TW / CB_PX;
    constexpr int HALO = (KW - 1 + 3) / 4 * 4;             // halo columns, rounded so that rows stay 16-B aligned
    constexpr int PITCH = CB_TW + (HALO < 8 ? 8 : HALO);
    static_assert(PITCH <= 192, "three 64-lane passes stage a tile row");
    const int trows = CB_TH + kH - 1, nrows = nIn * trows;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    struct Loc { int ent, grp, tile, x0, y0, H, W; };
    auto locate = [&](int idx) {
        Loc L;
        L.ent = 0;
        while (L.ent + 1 < cb.n && idx >= cb.blk0[L.ent + 1]) ++L.ent;         // (block-uniform; at most 10 entries)
        L.H = cb.H[L.ent]; L.W = cb.W[L.ent];
        const int tilesx = (L.W - KW + 1 + CB_TW - 1) / CB_TW;
        const int rel = idx - cb.blk0[L.ent];
        L.grp = rel % groups; L.tile = rel / groups;                           // (the groups of a tile next to each other: they share its input)
        const int by = L.tile / tilesx, bx = L.tile - by * tilesx;
        L.x0 = bx * CB_TW; L.y0 = by * CB_TH;
        return L;
    };
    // stage: every input plane's tile (clamped at the frame edge: those values only feed outputs that are not stored), by LDS-DMA -- a
    // wave takes whole tile rows (row = wave, wave + 4, ...), its lanes the row's columns in up to three requests, all in flight at once.
    // (Before: through registers, two rows' loads in flight before the first LDS write -- VGA learned pyramid 0.181 -> 0.175 ms, version2
    //  0.789 -> 0.767.  The very first version dealt single elements to threads: two integer divisions per element cost as many
    //  instructions as the convolution itself.)
    auto stage = [&](const Loc &L, float *buf) {
        const float *__restrict__ in = cb.in[L.ent];
        const unsigned c0 = 4u * (unsigned)min(L.x0 + lane, L.W - 1), c1 = 4u * (unsigned)min(L.x0 + min(lane + 64, PITCH - 1), L.W - 1),
                       c2 = 4u * (unsigned)min(L.x0 + min(lane + 128, PITCH - 1), L.W - 1);
        // (plane i, tile row r) of tile row rr = wave, wave + 4, ...: carried, not divided out (trows >= 8 > 4); the row's element
        // index in 32 bits (the launcher checks nIn * H * W)
        int i = 0, r = wv;
        for (int rr = wv; rr < nrows; rr += 4) {
            const unsigned e0 = (unsigned)i * cb.plane[L.ent] + (unsigned)min(L.y0 + r, L.H - 1) * cb.pitch[L.ent];
            cb_glds_row<PITCH>(c0, c1, c2, in + e0, buf + rr * PITCH);
            r += 4;
            if (r >= trows) { r -= trows; ++i; }
        }
    };
    // (Tried: blocks that work through 2 or 4 consecutive (tile, output group) items, the next item's tile requested into a second LDS
    //  buffer before the arithmetic of the current one -- with one item per block the 982 blocks of a VGA pyramid layer are all resident
    //  at once and run load, arithmetic and store in lockstep.  Slower everywhere: VGA learned pyramid 0.175 -> 0.196 (2) / 0.244 (4),
    //  version2 0.767 -> 1.225 / 1.54 ms: half / a quarter of the waves per CU hide less than the prefetch does.
    //  profiles/r04_at_conv_items_per_block.txt)
    const Loc L = locate(blockIdx.x);
    const float *cur = conv_smem;
    stage(L, conv_smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int tx = threadIdx.x % TXN, ty = threadIdx.x / TXN;
    {
        const int o0 = L.grp * NT;
        // Two pixels per instruction: the multiply and the add of a tap are v_pk_mul_f32 / v_pk_add_f32 on pixel pairs (the weight is an
        // SGPR pair with op_sel broadcasting its low half) -- each element rounded exactly as the scalar multiply and add are, in the same
        // order, so the results do not change; the odd taps' pixel pairs (v[k], v[k+1] with k odd: not an aligned register pair) are
        // copies made once per input row and shared by every output plane.  The scalar form issued every operation on the VALU's slow
        // path (SGPR operand: 0.9 per cycle and CU); this one issues half as many.
        static_assert(CB_PX == 4, "two pixel pairs per thread");
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 acc[NT][2];
#pragma unroll
        for (int o = 0; o < NT; ++o) {
            const float b = cb.bias[L.ent] ? cb.bias[L.ent][o0 + o] : 0.f;
            acc[o][0] = f2{b, b};
            acc[o][1] = f2{b, b};
        }
        const float *__restrict__ w = cb.w[L.ent];
        for (int i = 0; i < nIn; ++i)
            for (int u = 0; u < kH; ++u) {
                const float4 *row = reinterpret_cast<const float4 *>(cur + (i * trows + ty + u) * PITCH + CB_PX * tx);
                constexpr int NV4 = (CB_PX + KW - 1 + 3) / 4;                                  // 16-B pieces of the tile row this strip reads
                float v[4 * NV4];
#pragma unroll
                for (int j = 0; j < NV4; ++j) { const float4 tt = row[j]; v[4 * j] = tt.x; v[4 * j + 1] = tt.y; v[4 * j + 2] = tt.z; v[4 * j + 3] = tt.w; }
                f2 pr[KW][2];
#pragma unroll
                for (int k = 0; k < KW; ++k) { pr[k][0] = f2{v[k], v[k + 1]}; pr[k][1] = f2{v[k + 2], v[k + 3]}; }
#pragma unroll
                for (int o = 0; o < NT; ++o) {
                    typedef const float __attribute__((address_space(4))) *cfp;               // constant address space: the backend selects SMEM
                    const cfp wr = (cfp)(w + (((long long)(o0 + o) * nIn + i) * kH + u) * KW);   // (the index is wave-uniform)
#pragma unroll
                    for (int k = 0; k < KW; ++k) {
                        const float wk = wr[k];
                        const f2 w2 = f2{wk, wk};
                        acc[o][0] = acc[o][0] + w2 * pr[k][0];
                        acc[o][1] = acc[o][1] + w2 * pr[k][1];
                    }
                }
            }
        const int Ho = L.H - kH + 1, Wo = L.W - KW + 1;
        const int y = L.y0 + ty;
        if (y < Ho) {
            float *__restrict__ out = cb.out[L.ent];
#pragma unroll
            for (int o = 0; o < NT; ++o)
#pragma unroll
                for (int q = 0; q < CB_PX; ++q) {
                    const int x = L.x0 + CB_PX * tx + q;
                    const float a = acc[o][q >> 1][q & 1];
                    if (x < Wo) out[((long long)(o0 + o) * Ho + y) * Wo + x] = TANH ? tanhf(a) : a;
                }
        }
    }
}

template <int KW, int NT>
static bool launch_conv_batch(dfe_ctx *ctx, const ConvBatch &cb, int n, int nIn, int nOut, int kH, int tanh_after, int maxblocks) {
    const int groups = nOut / NT;
    constexpr int halo = (KW - 1 + 3) / 4 * 4;
    // tile shape (option conv_narrow forces): 64 x 16 for large kernels -- 17 x 17 stages 2.5 values per output instead of 3.4 and
    // version2's layer runs 0.594 -> 0.556 ms -- and 128 x 8 otherwise: at 5 x 5 the narrow tiles were 4 % SLOWER (VGA learned pyramid
    // 0.182 -> 0.189 ms, 1080p 0.86 -> 0.89) although they waste 9 % of their area at 645 columns instead of 19 %: a wave then spans
    // four tile rows and the block stages 20 rows per plane instead of 12 (profiles/r04_aq_conv_tile_shapes.txt).
    const bool narrow = ctx->opt_bool(DFE_OPT_CONV_NARROW, KW >= 9 && kH >= 9);
    const int TW = narrow ? 64 : 128, TH = narrow ? 16 : 8;
    const size_t lds = (size_t)nIn * (TH + kH - 1) * (TW + (halo < 8 ? 8 : halo)) * sizeof(float);
    if (lds > 64 * 1024) return false;
    auto kern = narrow ? (tanh_after ? conv_batch_kernel<KW, NT, true, true> : conv_batch_kernel<KW, NT, false, true>)
                       : (tanh_after ? conv_batch_kernel<KW, NT, true, false> : conv_batch_kernel<KW, NT, false, false>);
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    ConvBatch c2 = cb;
    c2.n = n;
    c2.blk0[0] = 0;
    for (int e = 0; e < n; ++e) c2.blk0[e + 1] = c2.blk0[e] + dfe_cdiv(cb.W[e] - KW + 1, TW) * dfe_cdiv(cb.H[e] - kH + 1, TH) * groups;
    (void)maxblocks;
    hipLaunchKernelGGL(kern, dim3(c2.blk0[n]), dim3(256), lds, ctx->stream, c2, nIn, nOut, kH, groups);
    return true;
}

// n inputs through ONE layer (full connection): in[e] [nIn][H[e]][W[e]] -> out[e]; per-entry weights (the scales may have their own).
// Falls back to one dfe_filter_layer_forward launch per entry for shapes without a batched instantiation.
static int conv_batch_try(dfe_ctx *ctx, int n, const float *const *in, const dfe_filter_layer *const *L, const int *H, const int *W, float *const *out, bool *done_out,
                          const int *in_pitch = nullptr, const long long *in_plane = nullptr) {
    *done_out = false;
    const dfe_filter_layer &L0 = *L[0];
    bool same = true;
    for (int e = 1; e < n; ++e)
        same = same && L[e]->nIn == L0.nIn && L[e]->nOut == L0.nOut && L[e]->kH == L0.kH && L[e]->kW == L0.kW && L[e]->tanh_after == L0.tanh_after && !L[e]->conn;
    if (same && !L0.conn && ctx->opt[DFE_OPT_CONV_BATCH] != 0) {
        ConvBatch cb;
        int maxblocks = 0;
        bool ok = true;
        for (int e = 0; e < n; ++e) {
            DFE_REQUIRE(ctx, in[e] && out[e] && L[e]->weight && H[e] >= L0.kH && W[e] >= L0.kW, DFE_E_SHAPE, "filter layer batch: entry %d: %dx%d kernel on %dx%d", e,
                        L0.kH, L0.kW, H[e], W[e]);
            const long long pit = in_pitch ? in_pitch[e] : W[e], pla = in_plane ? in_plane[e] : (long long)H[e] * W[e];
            if (pit < W[e] || pla < (long long)(H[e] - 1) * pit + W[e]) ok = false;
            if ((long long)L0.nIn * pla >= (1ll << 32)) ok = false;                // (32-bit element indices in the staging)
            cb.in[e] = in[e]; cb.out[e] = out[e]; cb.w[e] = L[e]->weight; cb.bias[e] = L[e]->bias; cb.H[e] = H[e]; cb.W[e] = W[e];
            cb.pitch[e] = (unsigned)pit; cb.plane[e] = (unsigned)pla;
            const int b = dfe_cdiv(W[e] - L0.kW + 1, 64) * dfe_cdiv(H[e] - L0.kH + 1, 8);
            if (b > maxblocks) maxblocks = b;
        }
        // output planes per thread: the most that still leaves two blocks per CU (a 320 x 180 frame is 66 tiles: with all of a layer's
        // planes in one block most of the chip idles -- tests/time_matching.lua's filter 0.131 -> 0.105 ms with 2 planes per thread)
        long long tiles = 0;
        const bool narrow_k = ctx->opt_bool(DFE_OPT_CONV_NARROW, L0.kW >= 9 && L0.kH >= 9);   // (the tile shape launch_conv_batch picks)
        for (int e = 0; e < n; ++e) tiles += (long long)dfe_cdiv(W[e] - L0.kW + 1, narrow_k ? 64 : 128) * dfe_cdiv(H[e] - L0.kH + 1, narrow_k ? 16 : 8);
        const int cand[6] = {10, 8, 5, 4, 2, 1};
        int nt = 0;
        for (int c = 0; c < 6; ++c) {
            if (L0.nOut % cand[c]) continue;
            if (cand[c] == 10 && ctx->opt[DFE_OPT_CONV_NT10] == 0) continue;
            if (L0.kW == 17 && (cand[c] == 10 || cand[c] == 5 || cand[c] == 1)) continue;     // (instantiated for 17 x 17: 8, 4, 2)
            nt = cand[c];
            if (tiles * (L0.nOut / nt) >= 2ll * ctx->ncu) break;
        }
        bool done = false;
#define DFE_CB(KWV)                                                                                                                       \
    if (L0.kW == KWV) {                                                                                                                   \
        if (nt == 8) done = launch_conv_batch<KWV, 8>(ctx, cb, n, L0.nIn, L0.nOut, L0.kH, L0.tanh_after, maxblocks);                      \
        else if (nt == 10) done = launch_conv_batch<KWV, 10>(ctx, cb, n, L0.nIn, L0.nOut, L0.kH, L0.tanh_after, maxblocks);               \
        else if (nt == 5) done = launch_conv_batch<KWV, 5>(ctx, cb, n, L0.nIn, L0.nOut, L0.kH, L0.tanh_after, maxblocks);                 \
        else if (nt == 4) done = launch_conv_batch<KWV, 4>(ctx, cb, n, L0.nIn, L0.nOut, L0.kH, L0.tanh_after, maxblocks);                 \
        else if (nt == 2) done = launch_conv_batch<KWV, 2>(ctx, cb, n, L0.nIn, L0.nOut, L0.kH, L0.tanh_after, maxblocks);                 \
        else if (nt == 1) done = launch_conv_batch<KWV, 1>(ctx, cb, n, L0.nIn, L0.nOut, L0.kH, L0.tanh_after, maxblocks);                 \
    }
        if (ok && nt) { DFE_CB(3) DFE_CB(5) DFE_CB(7) }
        if (ok && L0.kW == 17) {                                                                                   // version2/network.lua's 17 x 17 x 32
            if (nt == 8) done = launch_conv_batch<17, 8>(ctx, cb, n, L0.nIn, L0.nOut, L0.kH, L0.tanh_after, maxblocks);
            else if (nt == 4) done = launch_conv_batch<17, 4>(ctx, cb, n, L0.nIn, L0.nOut, L0.kH, L0.tanh_after, maxblocks);
            else if (nt == 2) done = launch_conv_batch<17, 2>(ctx, cb, n, L0.nIn, L0.nOut, L0.kH, L0.tanh_after, maxblocks);
        }
#undef DFE_CB
        if (done) {
            DFE_LAUNCH_CHECK(ctx);
            *done_out = true;
        }
    }
    return DFE_OK;
}

int dfe_filter_layer_forward_batch(dfe_ctx *ctx, int n, const float *const *in, const dfe_filter_layer *const *L, const int *H, const int *W,
                                   float *const *out) {
    DFE_REQUIRE(ctx, n >= 1 && n <= 2 * DFE_MAX_RATIOS, DFE_E_ARG, "filter layer batch: n=%d", n);
    bool done = false;
    int rc0 = conv_batch_try(ctx, n, in, L, H, W, out, &done);
    if (rc0 != DFE_OK || done) return rc0;
    for (int e = 0; e < n; ++e) {
        int rc = dfe_filter_layer_forward(ctx, in[e], *L[e], H[e], W[e], out[e]);
        if (rc) return rc;
    }
    return DFE_OK;
}

// the batched kernel on inputs that are VIEWS (rows in_pitch[e] floats apart, planes in_plane[e] floats apart): the single-scale model's
// first branch reads the part of frame 0 its narrowed features come from in place.  *done = false: no batched instantiation takes the
// layer (the caller makes the view contiguous and goes through dfe_filter_layer_forward_batch)
int dfe_filter_layer_forward_batch_view(dfe_ctx *ctx, int n, const float *const *in, const dfe_filter_layer *const *L, const int *H, const int *W, const int *in_pitch,
                                        const long long *in_plane, float *const *out, bool *done) {
    *done = false;
    DFE_REQUIRE(ctx, n >= 1 && n <= 2 * DFE_MAX_RATIOS, DFE_E_ARG, "filter layer batch: n=%d", n);
    return conv_batch_try(ctx, n, in, L, H, W, out, done, in_pitch, in_plane);
}

int dfe_filter_layer_forward(dfe_ctx *ctx, const float *in, const dfe_filter_layer &L, int H, int W, float *out) {
    DFE_REQUIRE(ctx, in && L.weight && out, DFE_E_ARG, "filter layer: NULL tensor");
    DFE_REQUIRE(ctx, L.nIn > 0 && L.nOut > 0 && L.kH > 0 && L.kW > 0 && H >= L.kH && W >= L.kW && (!L.conn || L.nConn > 0), DFE_E_SHAPE,
                "filter layer: %d->%d planes, %dx%d kernel on %dx%d", L.nIn, L.nOut, L.kH, L.kW, H, W);
    const int g = grid_n((long long)L.nOut * (H - L.kH + 1) * (W - L.kW + 1));
    if (L.tanh_after)
        hipLaunchKernelGGL(conv_layer_kernel<true>, dim3(g), dim3(256), 0, ctx->stream, in, L.weight, L.bias, (const int *)L.conn, L.nConn, L.nIn, L.nOut, H, W, L.kH, L.kW, out);
    else
        hipLaunchKernelGGL(conv_layer_kernel<false>, dim3(g), dim3(256), 0, ctx->stream, in, L.weight, L.bias, (const int *)L.conn, L.nConn, L.nIn, L.nOut, H, W, L.kH, L.kW, out);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

extern "C" {

int dfe_spatial_convolution_f32(dfe_ctx *ctx, const float *in, const float *weight, const float *bias, int nIn, int nOut, int H, int W,
                                int kH, int kW, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, in && weight && out, DFE_E_ARG, "dfe_spatial_convolution_f32: NULL tensor");
    DFE_REQUIRE(ctx, nIn > 0 && nOut > 0 && kH > 0 && kW > 0 && H >= kH && W >= kW, DFE_E_SHAPE,
                "dfe_spatial_convolution_f32: %d->%d planes, %dx%d kernel on %dx%d", nIn, nOut, kH, kW, H, W);
    // the LDS-tiled kernel where it has an instantiation (same accumulation order, separately rounded multiply and add: bit-identical
    // to conv_kernel -- 17 x 17, 3 -> 32 planes at VGA: 1.84 -> 0.36 ms)
    if (ctx->cv_mode != 1 && ctx->opt[DFE_OPT_CONV_BATCH] != 0) {
        dfe_filter_layer L{};
        L.weight = weight; L.bias = bias; L.nIn = nIn; L.nOut = nOut; L.kH = kH; L.kW = kW;
        const dfe_filter_layer *Lp = &L;
        bool done = false;
        int rc = conv_batch_try(ctx, 1, &in, &Lp, &H, &W, &out, &done);
        if (rc != DFE_OK || done) return rc;
    }
    hipLaunchKernelGGL(conv_kernel, dim3(grid_n((long long)nOut * (H - kH + 1) * (W - kW + 1))), dim3(256), 0, ctx->stream, in, weight, bias,
                       nIn, nOut, H, W, kH, kW, out);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

// nn.SpatialConvolution followed by nn.Tanh as one launch (getFilter puts a Tanh behind every layer but the last): the batched kernel's
// epilogue applies the same tanhf dfe_tanh_f32 does -- bit-identical to the two calls
int dfe_spatial_convolution_tanh_f32(dfe_ctx *ctx, const float *in, const float *weight, const float *bias, int nIn, int nOut, int H, int W, int kH, int kW,
                                     float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, in && weight && out, DFE_E_ARG, "dfe_spatial_convolution_tanh_f32: NULL tensor");
    DFE_REQUIRE(ctx, nIn > 0 && nOut > 0 && kH > 0 && kW > 0 && H >= kH && W >= kW, DFE_E_SHAPE,
                "dfe_spatial_convolution_tanh_f32: %d->%d planes, %dx%d kernel on %dx%d", nIn, nOut, kH, kW, H, W);
    dfe_filter_layer L{};
    L.weight = weight; L.bias = bias; L.nIn = nIn; L.nOut = nOut; L.kH = kH; L.kW = kW; L.tanh_after = 1;
    if (ctx->cv_mode != 1 && ctx->opt[DFE_OPT_CONV_BATCH] != 0) {
        const dfe_filter_layer *Lp = &L;
        bool done = false;
        int rc = conv_batch_try(ctx, 1, &in, &Lp, &H, &W, &out, &done);
        if (rc != DFE_OK || done) return rc;
    }
    return dfe_filter_layer_forward(ctx, in, L, H, W, out);
}

int dfe_spatial_convolution_map_f32(dfe_ctx *ctx, const float *in, const float *weight, const float *bias, const int32_t *conn, int nConn,
                                    int nIn, int nOut, int H, int W, int kH, int kW, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, in && weight && conn && out, DFE_E_ARG, "dfe_spatial_convolution_map_f32: NULL tensor");
    DFE_REQUIRE(ctx, nIn > 0 && nOut > 0 && nConn > 0 && kH > 0 && kW > 0 && H >= kH && W >= kW, DFE_E_SHAPE,
                "dfe_spatial_convolution_map_f32: %d connections %d->%d planes, %dx%d kernel on %dx%d", nConn, nIn, nOut, kH, kW, H, W);
    hipLaunchKernelGGL(conv_map_kernel, dim3(grid_n((long long)nOut * (H - kH + 1) * (W - kW + 1))), dim3(256), 0, ctx->stream, in, weight,
                       bias, conn, nConn, nOut, H, W, kH, kW, out);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_tanh_f32(dfe_ctx *ctx, const float *in, int64_t n, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, n >= 0, DFE_E_SHAPE, "dfe_tanh_f32: n=%lld", (long long)n);
    if (n == 0) return DFE_OK;
    DFE_REQUIRE(ctx, in && out, DFE_E_ARG, "dfe_tanh_f32: NULL tensor");
    hipLaunchKernelGGL(tanh_kernel, dim3(grid_n(n)), dim3(256), 0, ctx->stream, in, (long long)n, out);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// nn.SpatialContrastiveNormalization(nIn, 1-D kernel, threshold, thresval): version2/network.lua:12 puts it in front of both
// filter branches (image.gaussian1D(normalization_k)).  = SpatialSubtractiveNormalization then SpatialDivisiveNormalization,
// each built on one "estimator": zero pad, horizontal pass per plane, vertical pass that also sums the planes, divided by the
// estimator of a tensor of ones (border correction).  Un-vendored nn, nothing in the reference tests it: restated from recall
// (the CPU restatement states the same recall) -- parity unpinned.  Same term order as that restatement: bit-identical.
// ------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int CN_MAXK = 33;
struct CnKernel { float kn[CN_MAXK]; int k; };

// Two launches for up to two frames (blockIdx.z), each a tile kernel that does an estimator's horizontal pass, vertical pass and plane
// sum in LDS, in the term order of the stand-alone passes (`s = s + kn[v] * a` over v, then over planes and u: c outer, u inner; zero
// padding as operands, not as skipped terms):
//   MODE 0: est = estimator(in), coef = estimator(ones) (no memory operand: 17 + C * 17 terms per pixel, recomputed per call)
//   MODE 1: est2 = estimator((in - est / coef)^2), then out = (in - est / coef) / max-thresholded(sqrt(est2) / coef), optionally only a
//           crop window of it (version2's SpatialPadding(-lWin, -tWin, -rWin, -bWin) behind the first branch's normalisation).
// (First version: seven grid-stride launches per frame through a C-plane temporary -- 100 us per VGA frame, a fifth of version2's step.)
struct CnFused {
    const float *in[2];
    float *est[2], *coef[2], *out[2];
    int cx[2], cy[2], cw[2], ch[2];        // crop window of out[f] (cw x ch at (cx, cy); the whole frame: 0, 0, W, H)
    int C, H, W;
    float threshold, thresval;
    int coef_ready;                         // MODE 0: the coefficient plane (coef[0] == coef[1]) is already there
    CnKernel kk;
};
constexpr int CN_TW = 64, CN_TH = 16, CN_CG = 3;
extern __shared__ float cn_smem[];

template <int MODE, int KT>   // KT: the kernel size as a constant (taps unrolled, coefficients in scalar registers), 0 = any size <= CN_MAXK
__global__ __launch_bounds__(256) void cn_fused_kernel(CnFused a) {
#pragma clang fp contract(off)
    const int k = KT ? KT : a.kk.k, pl = k / 2, RH = CN_TH + k - 1, SW = CN_TW + k - 1;
    // up to CN_CG planes are staged and filtered together (round 5: one plane at a time meant three barriers per plane -- nine for an RGB
    // frame -- in a kernel whose 600 blocks are a launch of two rounds: 30 + 22 us per VGA pair, nearly all of it latency)
    float *src = cn_smem;                   // [CG][RH][SW] the planes' tiles with their halo, zero outside the frame
    float *tmpl = src;                      // [CG][RH][TW] after the horizontal pass: OVER the tiles (the pass goes through registers), so that
                                            // a block needs 31 / 41 KB and every block of a VGA pair is resident at once
    float *qt = src + CN_CG * RH * SW;      // [RH][SW] MODE 1: est / coef on the halo'd tile
    float *kl = qt + (MODE == 1 ? RH * SW : 0);   // KT == 0: the coefficients (a run-time index into the argument block would be a scalar load per tap)
    const int f = blockIdx.z, H = a.H, W = a.W, C = a.C;
    const float *in = a.in[f];
    const int x0 = blockIdx.x * CN_TW, y0 = blockIdx.y * CN_TH;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long P = (long long)H * W;
    auto kn = [&](int v) { return KT ? a.kk.kn[v] : kl[v]; };
    if (!KT) {
        if (threadIdx.x < CN_MAXK) kl[threadIdx.x] = a.kk.kn[threadIdx.x];
    }
    if (MODE == 1) {
        const float *est = a.est[f], *coef = a.coef[f];
        for (int i = threadIdx.x; i < RH * SW; i += 256) {
            const int r = i / SW, c = i - r * SW, yy = y0 + r - pl, xx = x0 + c - pl;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            qt[i] = ok ? est[(long long)yy * W + xx] / coef[(long long)yy * W + xx] : 0.f;
        }
    }
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < C; c0 += CN_CG) {
        const int nc = min(CN_CG, C - c0);
        __syncthreads();                    // the previous group's vertical pass is done with tmpl (and everyone with src)
        for (int i = threadIdx.x; i < RH * SW; i += 256) {
            const int r = i / SW, cc = i - r * SW, yy = y0 + r - pl, xx = x0 + cc - pl;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            float v[CN_CG];
#pragma unroll
            for (int g = 0; g < CN_CG; ++g) v[g] = (ok && g < nc) ? in[(c0 + g) * P + (long long)yy * W + xx] : 0.f;   // (all planes' loads in flight together)
#pragma unroll
            for (int g = 0; g < CN_CG; ++g) {
                if (g >= nc) break;
                float w = v[g];
                if (MODE == 1 && ok) { const float y = w - qt[i]; w = y * y; }      // (qt[i] is this thread's own write)
                src[g * RH * SW + i] = w;
            }
        }
        __syncthreads();
        constexpr int NR = (CN_TH + (KT ? KT : CN_MAXK) - 1 + 3) / 4;   // rows of the horizontal pass per thread
        float hp[CN_CG][NR];
#pragma unroll
        for (int g = 0; g < CN_CG; ++g)
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const int r = ty + 4 * n;
                float t = 0.f;
                if (g < nc && r < RH) {
                    const float *sr = src + g * RH * SW + r * SW + tx;
                    if (KT) {
#pragma unroll
                        for (int v = 0; v < (KT ? KT : 1); ++v) t = t + kn(v) * sr[v];
                    } else {
                        for (int v = 0; v < k; ++v) t = t + kn(v) * sr[v];
                    }
                }
                hp[g][n] = t;
            }
        __syncthreads();                    // everyone has read the tiles: their space takes the horizontal pass's results
#pragma unroll
        for (int g = 0; g < CN_CG; ++g)
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                const int r = ty + 4 * n;
                if (g < nc && r < RH) tmpl[g * RH * CN_TW + r * CN_TW + tx] = hp[g][n];
            }
        __syncthreads();
        for (int g = 0; g < nc; ++g) {      // (plane outer, tap inner: the term order of the stand-alone passes)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float *tc = tmpl + g * RH * CN_TW + (ty + 4 * j) * CN_TW + tx;
                if (KT) {
#pragma unroll
                    for (int u = 0; u < (KT ? KT : 1); ++u) s[j] = s[j] + kn(u) * tc[u * CN_TW];
                } else {
                    for (int u = 0; u < k; ++u) s[j] = s[j] + kn(u) * tc[u * CN_TW];
                }
            }
        }
    }
    const int x = x0 + tx;
    if (x >= W) return;
    if (MODE == 0) {
        float ro = 0.f;                     // the horizontal pass over a row of ones
        for (int v = 0; v < k; ++v) {
            const int xx = x + v - pl;
            ro = ro + kn(v) * ((xx >= 0 && xx < W) ? 1.f : 0.f);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = y0 + ty + 4 * j;
            if (y >= H) continue;
            a.est[f][(long long)y * W + x] = s[j];
            if (a.coef_ready || f != 0) continue;       // (one plane for both frames, kept in the ctx from call to call)
            float cf = 0.f;
            for (int c = 0; c < C; ++c)
                for (int u = 0; u < k; ++u) {
                    const int yy = y + u - pl;
                    cf = cf + kn(u) * ((yy >= 0 && yy < H) ? ro : 0.f);
                }
            a.coef[f][(long long)y * W + x] = cf;
        }
    } else {
        const int cx = a.cx[f], cy = a.cy[f], cw = a.cw[f], ch = a.ch[f];
        if (x < cx || x >= cx + cw) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = y0 + ty + 4 * j;
            if (y >= H || y < cy || y >= cy + ch) continue;
            const long long p = (long long)y * W + x;
            const float cf = a.coef[f][p];
            float sd = sqrtf(s[j]) / cf;
            sd = sd > a.threshold ? sd : a.thresval;
            const float q = a.est[f][p] / cf;
            for (int c = 0; c < C; ++c) a.out[f][((long long)c * ch + (y - cy)) * cw + (x - cx)] = (in[c * P + p] - q) / sd;
        }
    }
}

}  // namespace

// Normalises in0 (and in1 when not NULL) -- same C x H x W -- into out0 / out1; out0 receives only the crop window (cw x ch at (cx, cy))
// when cw > 0.  scratch: 4 * H * W floats of device memory.
int dfe_contrastive_normalization_run2(dfe_ctx *ctx, const float *in0, const float *in1, int C, int H, int W, const float *kernel_host, int k,
                                       float threshold, float thresval, float *scratch, float *out0, float *out1, int cx, int cy, int cw, int ch) {
    DFE_REQUIRE(ctx, in0 && kernel_host && out0 && scratch && (!in1 || out1), DFE_E_ARG, "dfe_contrastive_normalization_f32: NULL argument");
    DFE_REQUIRE(ctx, C > 0 && H > 0 && W > 0 && k > 0 && k <= CN_MAXK, DFE_E_SHAPE, "dfe_contrastive_normalization_f32: C=%d %dx%d kernel %d (max %d)", C, H,
                W, k, CN_MAXK);
    DFE_REQUIRE(ctx, cw <= 0 || (cx >= 0 && cy >= 0 && ch > 0 && cx + cw <= W && cy + ch <= H), DFE_E_SHAPE,
                "dfe_contrastive_normalization_f32: crop %dx%d at (%d, %d) of %dx%d", cw, ch, cx, cy, W, H);
    CnFused a{};
    a.kk.k = k;
    float ks = 0.f;
    for (int i = 0; i < k; ++i) ks += kernel_host[i];
    for (int i = 0; i < k; ++i) a.kk.kn[i] = kernel_host[i] / (ks * (float)C);     // self.kernel:div(self.kernel:sum() * self.nInputPlane)
    const long long P = (long long)H * W;
    const int nf = in1 ? 2 : 1;
    a.in[0] = in0; a.in[1] = in1; a.out[0] = out0; a.out[1] = out1;
    // the border-correction plane depends on (H, W, C, kernel) only: computed by the first call with these, read by every later one
    bool same = ctx->cn_coef && ctx->cn_key[0] == H && ctx->cn_key[1] == W && ctx->cn_key[2] == C && ctx->cn_key[3] == k;
    for (int i = 0; same && i < k; ++i) same = ctx->cn_key_kn[i] == a.kk.kn[i];
    if (!same) {
        if ((size_t)P > ctx->cn_coef_floats) {
            DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->cn_coef) DFE_HIP(ctx, hipFree(ctx->cn_coef));
            ctx->cn_coef = nullptr; ctx->cn_coef_floats = 0; ctx->cn_key[3] = 0;
            hipError_t e = hipMalloc(&ctx->cn_coef, (size_t)P * sizeof(float));
            if (e != hipSuccess) return dfe_fail(ctx, DFE_E_ALLOC, "normalisation coefficients hipMalloc(%lld floats): %s", P, hipGetErrorString(e));
            ctx->cn_coef_floats = (size_t)P;
        }
        ctx->cn_key[0] = H; ctx->cn_key[1] = W; ctx->cn_key[2] = C; ctx->cn_key[3] = k;
        for (int i = 0; i < k; ++i) ctx->cn_key_kn[i] = a.kk.kn[i];
    }
    a.coef_ready = same ? 1 : 0;
    for (int f = 0; f < 2; ++f) {
        a.coef[f] = ctx->cn_coef;
        a.est[f] = scratch + (2 * f + 1) * P;
        a.cx[f] = 0; a.cy[f] = 0; a.cw[f] = W; a.ch[f] = H;
    }
    if (cw > 0) { a.cx[0] = cx; a.cy[0] = cy; a.cw[0] = cw; a.ch[0] = ch; }
    a.C = C; a.H = H; a.W = W; a.threshold = threshold; a.thresval = thresval;
    const dim3 grid(dfe_cdiv(W, CN_TW), dfe_cdiv(H, CN_TH), nf);
    const int RH = CN_TH + k - 1, SW = CN_TW + k - 1;
    const size_t lds0 = ((size_t)CN_CG * RH * SW + CN_MAXK) * sizeof(float), lds1 = lds0 + (size_t)RH * SW * sizeof(float);
    void (*k0)(CnFused) = k == 17 ? cn_fused_kernel<0, 17> : cn_fused_kernel<0, 0>;
    void (*k1)(CnFused) = k == 17 ? cn_fused_kernel<1, 17> : cn_fused_kernel<1, 0>;
    if (lds0 > 64 * 1024) DFE_HIP(ctx, hipFuncSetAttribute((const void *)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0));
    if (lds1 > 64 * 1024) DFE_HIP(ctx, hipFuncSetAttribute((const void *)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    hipLaunchKernelGGL(k0, grid, dim3(256), lds0, ctx->stream, a);
    hipLaunchKernelGGL(k1, grid, dim3(256), lds1, ctx->stream, a);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

// scratch: (C + 3) * H * W floats of device memory (>= the 4 * H * W of the fused form)
int dfe_contrastive_normalization_run(dfe_ctx *ctx, const float *in, int C, int H, int W, const float *kernel_host, int k, float threshold,
                                      float thresval, float *scratch, float *out) {
    return dfe_contrastive_normalization_run2(ctx, in, nullptr, C, H, W, kernel_host, k, threshold, thresval, scratch, out, nullptr, 0, 0, 0, 0);
}

extern "C" int dfe_contrastive_normalization_f32(dfe_ctx *ctx, const float *in, int C, int H, int W, const float *kernel_host, int k, float threshold,
                                                 float thresval, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, in && kernel_host && out, DFE_E_ARG, "dfe_contrastive_normalization_f32: NULL argument");
    DFE_REQUIRE(ctx, C > 0 && H > 0 && W > 0, DFE_E_SHAPE, "dfe_contrastive_normalization_f32: C=%d %dx%d", C, H, W);
    void *scr = nullptr;
    int rc = dfe_scratch(ctx, ((size_t)C + 3) * H * W * sizeof(float), &scr);
    if (rc) return rc;
    return dfe_contrastive_normalization_run(ctx, in, C, H, W, kernel_host, k, threshold, thresval, (float *)scr, out);
}
