// ssd_cost_volume.hip -- A0+A1 / A1 / A1r: the dense SSD cost-volume build for gfx950.
//
// Two kernels produce out[y][x][dy][dx] = sum_k (a_k(y,x) - b_k(y+dy,x+dx))^2 in the reference's
// pixel-major layout (nn.SpatialMatching output H1 x W1 x maxh x maxw,
// opticalflow_model.lua:93-98, radial/radial_opticalflow_groundtruth.lua:83-87):
//
//  ssd_cv_ref_kernel    any C / kernel / window / feature maps.  One thread per output element,
//                       features summed in the reference order (c, i, j) with separately rounded
//                       multiply and add, so it is bit-identical to the CPU loop nest for any
//                       float input.  O(C*kh*kw) flops per output: correct everywhere, fast nowhere.
//
//  ssd_cv_tiled_kernel  the HBM-bound kernel (raw frames, C in {1,3}, square kernel K).
//                       The patch sum is a K x K box filter of the per-displacement squared-
//                       difference plane e_d(u,v) = sum_c (I0[c][u][v] - I1[c][u+dy-oy][v+dx-ox])^2,
//                       evaluated separably (horizontal K-sum, then a K-row register ring), so an
//                       output costs ~25 VALU ops instead of 3*C*K*K.
//                         * lane <-> displacement d = dy*wWin+dx: the 64 lanes of a wave own 64
//                           consecutive cells of one pixel's window, so every global store is a
//                           full-wave 256-B contiguous burst in the reference layout and the
//                           volume is written exactly once, with no LDS transpose;
//                         * frame-0 values are wave-uniform -> scalar (SMEM) loads, zero VALU/LDS;
//                         * the frame-1 tile (+ search halo) is staged once per block in LDS as
//                           channel-interleaved pixels; consecutive lanes read consecutive pixels
//                           (row pitch == wWin mod 16 keeps a wave that spans several dy rows
//                           conflict-free), one ds_read_b128 per squared difference;
//                         * edge tiles are shifted inwards instead of masked, so no lane ever
//                           reads out of bounds; duplicate pixels are not stored twice.
//                       Summation order is fixed relative to the output pixel (independent of the
//                       tile), so results do not depend on the launch geometry.
#include "dfe_internal.h"
#include <type_traits>
#include <memory>

// ------------------------------------------------------------------------------------------
// reference-order kernel
// ------------------------------------------------------------------------------------------
struct CvRefArgs {
    const float *a; long long a_plane; int a_pitch; int a_oy, a_ox;   // frame0 / in1
    const float *b; long long b_plane; int b_pitch;                   // frame1 / in2
    int C, kh, kw, Wo, hWin, wWin;
    long long total;                                                  // Ho*Wo*hWin*wWin
    float *out;
};

__global__ __launch_bounds__(256) void ssd_cv_ref_kernel(CvRefArgs p) {
#pragma clang fp contract(off)
    const int D = p.hWin * p.wWin;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < p.total; e += (long long)gridDim.x * 256) {
        long long pix = e / D;
        int d = (int)(e - pix * D);
        int y = (int)(pix / p.Wo), x = (int)(pix - (long long)y * p.Wo);
        int dy = d / p.wWin, dx = d - dy * p.wWin;
        float s = 0.f;
        for (int c = 0; c < p.C; ++c)
            for (int i = 0; i < p.kh; ++i) {
                const float *ap = p.a + c * p.a_plane + (long long)(y + p.a_oy + i) * p.a_pitch + x + p.a_ox;
                const float *bp = p.b + c * p.b_plane + (long long)(y + dy + i) * p.b_pitch + x + dx;
                for (int j = 0; j < p.kw; ++j) {
                    float t = ap[j] - bp[j];
                    float t2 = t * t;
                    s = s + t2;
                }
            }
        p.out[e] = s;
    }
}

static int launch_cv_ref(dfe_ctx *ctx, const CvRefArgs &a) {
    if (a.total <= 0) return DFE_OK;
    long long blocks = (a.total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    {
        DfeProfScope prof(ctx);
        hipLaunchKernelGGL(ssd_cv_ref_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
    }
    DFE_LAUNCH_CHECK(ctx);
    ctx->last_kernel = "ssd_cv_ref_kernel";
    return DFE_OK;
}

// ------------------------------------------------------------------------------------------
// tiled kernel
// ------------------------------------------------------------------------------------------
template <int C> struct Px;
template <> struct Px<1> { using type = float;  static constexpr int bank_mod = 32; };
template <> struct Px<3> { using type = float4; static constexpr int bank_mod = 16; };

template <int C> __device__ __forceinline__ float sqdiff(const float (&a)[C], const typename Px<C>::type &b);
template <> __device__ __forceinline__ float sqdiff<1>(const float (&a)[1], const float &b) {
    float d = a[0] - b;
    return d * d;
}
template <> __device__ __forceinline__ float sqdiff<3>(const float (&a)[3], const float4 &b) {
    float d0 = a[0] - b.x, d1 = a[1] - b.y, d2 = a[2] - b.z;
    return fmaf(d2, d2, fmaf(d1, d1, d0 * d0));
}

// horizontal K-sum of e[0..TX+K-2] -> h[0..TX-1]; fixed association relative to x.
template <int K, int TX> __device__ __forceinline__ void hsum(const float (&e)[TX + K - 1], float (&h)[TX]) {
    if constexpr (K == 7) {
        float p2[TX + 5], p4[TX + 3];
#pragma unroll
        for (int s = 0; s < TX + 5; ++s) p2[s] = e[s] + e[s + 1];
#pragma unroll
        for (int s = 0; s < TX + 3; ++s) p4[s] = p2[s] + p2[s + 2];
#pragma unroll
        for (int x = 0; x < TX; ++x) h[x] = (p4[x] + p2[x + 4]) + e[x + 6];
    } else {
#pragma unroll
        for (int x = 0; x < TX; ++x) {
            float s = e[x];
#pragma unroll
            for (int j = 1; j < K; ++j) s += e[x + j];
            h[x] = s;
        }
    }
}

// Row-image kernel: the same K-sums in block form (van Herk): the window of K = 7 over TX = 8 columns spans exactly two
// blocks of 7 positions, so h[x] = (suffix sum of block A from x) + (prefix sum of block B up to x-1): 18 adds per 8 outputs
// instead of 40.  Only non-negative terms are added (no running-sum cancellation), but the association depends on the
// column's position in its tile -- the row-image kernel therefore stores every pixel from exactly one tile (the shifted last
// tile skips the columns it shares with its neighbour), so results stay deterministic and independent of the schedule; on
// float frames they differ from the tiled kernel's by reassociation (integer-valued frames: exact either way).
template <int K, int TX> __device__ __forceinline__ void hsum_vh(const float (&e)[TX + K - 1], float (&h)[TX]) {
    if constexpr (K == 7 && TX == 8) {
        float sa[7], pb[7];
        sa[6] = e[6];
#pragma unroll
        for (int i = 5; i >= 0; --i) sa[i] = e[i] + sa[i + 1];
        pb[0] = e[7];
#pragma unroll
        for (int j = 1; j < 7; ++j) pb[j] = pb[j - 1] + e[7 + j];
        h[0] = sa[0];
#pragma unroll
        for (int x = 1; x < 7; ++x) h[x] = sa[x] + pb[x - 1];
        h[7] = pb[6];
    } else if constexpr (K == 7 && TX == 2) {
        const float s = ((e[1] + e[2]) + (e[3] + e[4])) + (e[5] + e[6]);
        h[0] = e[0] + s;
        h[1] = s + e[7];
    } else {
        hsum<K, TX>(e, h);
    }
}

// Wave-uniform loads of N consecutive floats through the scalar cache (s_load_dwordx8/x4/x2/x1).
// The constant address space makes the backend select SMEM; dword alignment is all SMEM needs.
typedef const float __attribute__((address_space(4))) *cfptr;
typedef float f8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef f8_t f8u_t __attribute__((aligned(4)));
typedef f4_t f4u_t __attribute__((aligned(4)));
typedef f2_t f2u_t __attribute__((aligned(4)));
template <int N, int OFF = 0> __device__ __forceinline__ void uload(cfptr p, float *v) {
    if constexpr (N >= 8) {
        f8_t t = *(const __attribute__((address_space(4))) f8u_t *)(p + OFF);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[OFF + i] = t[i];
        uload<N - 8, OFF + 8>(p, v);
    } else if constexpr (N >= 4) {
        f4_t t = *(const __attribute__((address_space(4))) f4u_t *)(p + OFF);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[OFF + i] = t[i];
        uload<N - 4, OFF + 4>(p, v);
    } else if constexpr (N >= 2) {
        f2_t t = *(const __attribute__((address_space(4))) f2u_t *)(p + OFF);
        v[OFF] = t[0];
        v[OFF + 1] = t[1];
        uload<N - 2, OFF + 2>(p, v);
    } else if constexpr (N == 1) {
        v[OFF] = p[OFF];
    }
}

struct CvTiledArgs {
    int H, W, hWin, wWin, Ho, Wo;
    long long plane;   // channel plane stride in elements (H*W of the full frame, also for row bands)
    int pitch;         // LDS row pitch in pixels
    int lrows;         // LDS rows = ROWS + hWin - 1
    int lcols;         // staged columns = TX + K - 1 + wWin - 1
    int chunk0;        // tiled kernel: first 64-displacement chunk it covers (0 = all)
    int seg_rows;      // row-image kernel: output rows per block (static tile height / column-sweep segment)
    int tile0_off;     // row-image kernel: byte offset of the frame-0 tile inside dynamic LDS
    int stage_off;     // row-image kernel: byte offset of the run images inside dynamic LDS
    int stage_len;     // row-image kernel: floats per image
    int sw_ovh;        // column sweep: cost of starting a piece, in row steps (warm-up rows + ring staging)
    int sw_min;        // column sweep: a piece is never shorter than this many output rows
    float scale;       // fp16 volume: stored value = cost * scale (2^-8 keeps 147 * 255^2 inside the half range)
    int f16 = 0;       // tiled multi kernel (pyramid volumes): out is a half volume, out[..] = half(cost * scale)
};

// Persistent column sweep: the (column, output row) grid is walked column-major as one linear sequence of ncols*Ho row
// steps and cut into B contiguous ranges, one per block.  A range is swept as pieces -- one per column it touches -- and
// every piece pays `ovh` row steps (K-1 warm-up rows + staging the rings), so the cuts are taken uniformly in the
// weighted coordinate u = pos + ovh * column(pos): every block then has the same rows + pieces*ovh.  Cuts closer than
// `minr` rows to a column's first or last row snap to the column boundary (no piece shorter than minr).
// (32-bit arithmetic: the launcher checks (Ho + ovh) * ncols * B < 2^31)
__host__ __device__ inline int sweep_cut(int b, int B, int ncols, int Ho, int ovh, int minr) {
    if (b >= B) return ncols * Ho;
    const int Lc = Ho + ovh, u = Lc * ncols * b / B;
    int col = u / Lc;
    int row = u - col * Lc - ovh;
    if (row < minr) row = 0;
    else if (Ho - row < minr) { ++col; row = 0; }
    return col * Ho + row;
}

extern __shared__ __attribute__((aligned(16))) char dfe_smem[];

#ifndef DFE_RI_SMEM
#define DFE_RI_SMEM true   // row-image kernel: frame-0 values through warmed scalar loads (true) or LDS + DPP (false)
#endif
#ifndef DFE_A32
#define DFE_A32 1
#endif
#ifndef DFE_NQW
// waves that share the 17th chunk of a 33 x 33 window.  Measured at VGA: 8 waves x 1 column (shorter critical path, but 7
// squared differences per output instead of 4) 0.3124 / 0.2529 ms fused / unfused against 0.3079 / 0.2407 ms for 4 x 2;
// 2 waves x 4 columns spill (24 registers of ring state).
#define DFE_NQW 4
#endif
#define DFE_CV_KERNEL_REV "cv-r4.1"
#ifndef DFE_SMEM_JIT
#define DFE_SMEM_JIT 0   // tuning: frame-0 scalars of a row loaded at its start instead of one row ahead
#endif
// -DDFE_MARKERS=1 (tools/isa_regions.py, no product build): assembler comments at the phase boundaries of the row loop, so that the ISA
// between them can be counted by class (VALU / v_readlane / LDS / ...) per phase
#if defined(DFE_MARKERS) && DFE_MARKERS
#define DFE_MARK(name) asm volatile("; DFE_MARK " name)
#else
#define DFE_MARK(name) do { } while (0)
#endif
#ifndef DFE_LEAD_FROM_IMAGE
#define DFE_LEAD_FROM_IMAGE 1
#endif
#ifndef DFE_Q_UNDEF
#define DFE_Q_UNDEF 1
#endif
#ifndef DFE_ROLES_STATIC
#define DFE_ROLES_STATIC 1
#endif
#ifndef DFE_ROLES
#define DFE_ROLES 1
#endif
#ifndef DFE_REFILL_AHEAD
#define DFE_REFILL_AHEAD 2   // column sweep: row steps between the request of a ring row and its deposit in LDS (1, 2 or 3)
#endif
#ifndef DFE_LW_PRIO
#define DFE_LW_PRIO 0        // column sweep: issue priority of the wave that refills the rings
#endif
#ifndef DFE_SCAN_AFTER_COPY
#define DFE_SCAN_AFTER_COPY 0   // tuning: the fused arg-min scan behind the copy-out instead of in front of it
#endif
#ifndef DFE_SCAN_SIDX
#define DFE_SCAN_SIDX 0   // fused scan: the winning lane's first cell found on the scalar unit (1) or by per-lane select chains + v_readlane (0).
                          // Measured in one call (r04_c): 1 is 0.5 % SLOWER at VGA and 0.3 % at 1080p -- 8 fewer vector instructions, but a serial
                          // chain of 8 compare -> scalar test pairs on the scan waves' path to the barrier
#endif
#ifndef DFE_COPY_FAST
#define DFE_COPY_FAST 1   // copy-out of whole runs: unclamped pieces addressed by one lane offset + scalar bases (0: every piece clamped and compared)
#endif
#ifndef DFE_CW0
#define DFE_CW0 5    // row-image kernel: first wave that takes part in the copy-out (0 = all waves)
#endif
#ifndef DFE_REC_ST_FLAGS
#define DFE_REC_ST_FLAGS ""   // cache-policy bits of the per-pixel record stores (tuning)
#endif
#ifndef DFE_ST_FLAGS
// cache-policy bits of the copy-out stores.  The volume streams out and nothing re-reads it from L2: with the non-temporal
// hint the build measures 268 instead of 285 us at VGA and the step's finalize pass finds its planes still cached
// (step -4 %); " sc1", " sc0 sc1" and combinations with " nt" measure the same as " nt" alone.  (Whole lines only: on the
// tiled kernel's 256-B per-wave dword stores the same hint costs 44 % -- 554 against 385 us.)
#define DFE_ST_FLAGS " nt"
#endif


// global_store_dword with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset:
// no VALU address arithmetic per store.  Nothing in the kernel reads what it stores, and stores
// need no wait before s_endpgm, so the compiler's vmcnt bookkeeping is not involved.
__device__ __forceinline__ void store_uniform_base(const void *base, unsigned lane_bytes, float v) {
    asm volatile("global_store_dword %0, %1, %2" ::"v"(lane_bytes), "v"(v), "s"(base) : "memory");
}

// Fused flow epilogue of one task row (TX columns x 64 cells of chunk `chunk`, pixel pg0 = column 0): leaves, per column,
// the chunk minimum and the 0-based index of the first cell attaining it in fa.part, plus the centre cell's cost and the
// pixel's first DFE_LEAD cells.  Costs are >= 0, so their bit patterns order like integers.  The butterfly leaves lane L
// with the chunk minimum of column L&7 (idle lanes carry +inf).  The first cell attaining it: compare every lane's value
// with the column minimum (a v_cmp IS a ballot), take the lowest set bit on the scalar unit and drop it into lane x.
// the part of the fused epilogue that is plain stores: the centre cell's cost and the pixel's first DFE_LEAD cells
template <int TX>
__device__ __forceinline__ void fuse_plain_stores(const float (&vrow)[TX], int lane, int chunk, long long pg0, const CvFuseArgs &fa, int nover,
                                                  int cmid, int lmid) {
    // Both are SGPR base + 32-bit lane offset dwordx4 stores: with 64-bit per-lane addresses the address arithmetic of these
    // few bytes cost the two waves that own them ~6 % of the whole fused row-image kernel (they sit before the barrier).
    static_assert(TX == 8, "two dwordx4 stores per lane");
    const f4_t lo = {vrow[0], vrow[1], vrow[2], vrow[3]}, hi = {vrow[4], vrow[5], vrow[6], vrow[7]};
    // (row-image kernel, shifted last tile: the first nover columns are the neighbour's and are not stored from here.  That
    //  rare path takes its column count through an asm barrier so that its per-column masks cannot be hoisted out of the
    //  caller's row loop -- eight SGPR pairs held across the sweep, as first written, cost every tile ~30 v_readlane per row)
    if (chunk == cmid && lane == lmid) {      // the lane that owns the centre cell: 8 pixels = 32 B
        const float *cb = fa.centre + pg0;
        if (nover == 0) {   // (wave-uniform)
            asm volatile("global_store_dwordx4 %0, %1, %2\n\tglobal_store_dwordx4 %0, %3, %2 offset:16" ::"v"(0u), "v"(lo), "s"(cb), "v"(hi) : "memory");
        } else {
            int nv = nover;
            asm volatile("" : "+s"(nv));
#pragma unroll
            for (int x = 0; x < TX; ++x)
                if (x >= nv) asm volatile("global_store_dword %0, %1, %2 offset:%3" ::"v"(0u), "v"(vrow[x]), "s"(cb), "n"(x * 4) : "memory");
        }
    }
    if (chunk == 0 && lane < DFE_LEAD) {
        // the pixel's first cells, for extractOutput, pixel-major [P][DFE_LEAD]: the 8 pixels of the tile row are 512
        // contiguous, line-aligned bytes, written back to back by this wave (8 x 64 B) so that they leave L2 as whole
        // lines.  (Cell-major planes -- 2 dwordx4 stores -- left 16 partial lines per row step, 690 k per VGA launch, and
        // partial lines that miss L2 are what this memory system is slow at: section 4.2.)
        const float *lb = fa.lead + pg0 * DFE_LEAD;
        const unsigned off = (unsigned)lane * 4u;
        if (nover == 0) {
#pragma unroll
            for (int x = 0; x < TX; ++x)
                asm volatile("global_store_dword %0, %1, %2 offset:%3" ::"v"(off), "v"(vrow[x]), "s"(lb), "n"(x * DFE_LEAD * 4) : "memory");
        } else {
            int nv = nover;
            asm volatile("" : "+s"(nv));
#pragma unroll
            for (int x = 0; x < TX; ++x)
                if (x >= nv) asm volatile("global_store_dword %0, %1, %2 offset:%3" ::"v"(off), "v"(vrow[x]), "s"(lb), "n"(x * DFE_LEAD * 4) : "memory");
        }
    }
}

template <int TX>
__device__ __forceinline__ void fuse_epilogue(const float (&vrow)[TX], bool valid, int lane, int chunk, long long pg0,
                                              const CvFuseArgs &fa) {
    // (plain stores first: after them the values are only needed as keys)
    fuse_plain_stores<TX>(vrow, lane, chunk, pg0, fa, 0, fa.cmid, fa.lmid);
    __builtin_amdgcn_sched_barrier(0);
    int key[TX];
#pragma unroll
    for (int x = 0; x < TX; ++x) key[x] = valid ? __float_as_int(vrow[x]) : 0x7f800000;
    const int wk = wave_min8<TX>(key, lane);
    int wl = 0;
    {
#define DFE_FIRST(x)                                                                                     \
    {                                                                                                    \
        const int f = __builtin_ctzll(__builtin_amdgcn_ballot_w64(key[x] == __builtin_amdgcn_readlane(wk, x))); \
        asm("v_writelane_b32 %0, %1, " #x : "+v"(wl) : "s"(f));                                          \
    }
        DFE_FIRST(0) DFE_FIRST(1) DFE_FIRST(2) DFE_FIRST(3) DFE_FIRST(4) DFE_FIRST(5) DFE_FIRST(6) DFE_FIRST(7)
#undef DFE_FIRST
    }
    if (lane < TX)
        fa.part[(long long)chunk * fa.Ptot + pg0 + lane] = make_float2(__int_as_float(wk), __int_as_float(chunk * 64 + wl));
}

// wave minimum of ONE value per lane: a plain 6-step reduction, all lanes get the result
__device__ __forceinline__ int wave_min1(int c) {
    c = min(c, __builtin_amdgcn_update_dpp(0, c, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    c = min(c, __builtin_amdgcn_update_dpp(0, c, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    c = min(c, __builtin_amdgcn_update_dpp(0, c, 0x124, 0xf, 0xf, false));   // row_ror:4
    c = min(c, __builtin_amdgcn_update_dpp(0, c, 0x128, 0xf, 0xf, false));   // row_ror:8
    const auto r = __builtin_amdgcn_permlane16_swap(c, c, false, false);
    c = min((int)r[0], (int)r[1]);
    const auto q = __builtin_amdgcn_permlane32_swap(c, c, false, false);
    return min((int)q[0], (int)q[1]);
}

// (wave_reduce8_transposed and fine_epilogue -- the fused pyramid scales' epilogue, shared with the feature matcher -- live in dfe_internal.h)

// Rows are swept in groups of U (the unroll that makes every ring index static):
//   K == 7: U = 6, vertical sum as the fixed tree ((H0+H1)+(H2+H3))+((H4+H5)+H6) kept as a ring of six
//           pair sums P_r = H_r + H_{r+1}  -> 4 adds per output;
//   else  : U = K, ring of K horizontal sums added oldest-first -> K-1 adds per output.
// Either way the association is fixed relative to the output pixel, so results do not depend on tiling.
template <int K> struct VUnroll { static constexpr int value = (K == 7) ? 6 : K; };

// A block owns NT horizontally adjacent TX-pixel tiles (one staged frame-1 tile for all of them) and
// TY = U*NQ-(K-1) output rows; its NT*nchunks (tile, 64-displacement chunk) tasks are dealt to NW waves,
// which then run independently (no barrier after the staging one).
// FUSE: the kernel also leaves, per (chunk, pixel), the chunk's minimum cost + index, the centre cell's cost and (from
// the wave that holds chunk 0) the extractOutput score, so the flow needs no second pass over the 1.16 GB volume: the
// values are still in registers when they are stored.  The wave minimum is a 6-step DPP reduction, the first-index
// rule a ballot; the arithmetic added (~50 %) hides under the store stream that bounds this kernel.
// SOFT (windows of at most one chunk): where `prob` is given, a task row leaves as soft-min probabilities instead of costs --
// p = e / sum(e), e = expf(-c - max(-c)) over the cells of a pixel = the lanes of the wave, with the arithmetic of
// softmin_kernel (multiscale.hip), so the result is bit-identical to running that kernel on the stored volume.
// FINE (8 x 8 windows, the finest scale of the multiscale matcher): nothing is stored; a task row goes through fine_epilogue below
template <int C, int K, int TX, int NT, int NW, int NQ, bool FUSE, bool SOFT = false, bool FINE = false, bool FINE16 = false, bool MID = false>
__device__ __forceinline__ void ssd_cv_tiled_body(const float *__restrict__ I0, const float *__restrict__ I1, float *__restrict__ out,
                                                  const CvTiledArgs &p, const CvFuseArgs &fa, int bx, int by, float *__restrict__ prob = nullptr,
                                                  const CvFineArgs *fine = nullptr) {
    using px_t = typename Px<C>::type;
    constexpr int U = VUnroll<K>::value;
    constexpr int ROWS = U * NQ;
    constexpr int TY = ROWS - (K - 1);
    constexpr int GX = NT * TX;
    constexpr int NE = TX + K - 1;
    px_t *lds = reinterpret_cast<px_t *>(dfe_smem);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0n = bx * GX, y0n = by * TY;                           // nominal origin
    const int x0 = min(x0n, p.Wo - GX), y0 = min(y0n, p.Ho - TY);      // shifted inwards at the frame edge
    const long long HW = p.plane;

    // ---- stage the frame-1 tile: rows y0.., cols x0.. (always inside the frame, see header) ----
    for (int r = wave; r < p.lrows; r += NW) {
        const float *src = I1 + (long long)(y0 + r) * p.W + x0;
        for (int s = lane; s < p.lcols; s += 64) {
            if constexpr (C == 1) {
                lds[r * p.pitch + s] = src[s];
            } else {
                lds[r * p.pitch + s] = make_float4(src[s], src[HW + s], src[2 * HW + s], 0.f);
            }
        }
    }
    __syncthreads();

    const int D = p.hWin * p.wWin;
    const int nchunks = (D + 63) >> 6;
    const int ncover = nchunks - p.chunk0;                       // chunks this launch covers
    const int ntasks = NT * ncover;
    const int oy = (p.hWin - 1) >> 1, ox = (p.wWin - 1) >> 1;

    // (Keeping sibling waves on the same row with an s_barrier per row was tried -- adjacent pieces then reach
    // L2 closer together -- and measured 4 % slower end to end; the waves run free.)
    for (int task = wave; task < ntasks; task += NW) {
        const int tile = task / ncover;                          // wave-uniform
        const int chunk = p.chunk0 + task - tile * ncover;
        const int xt = x0 + tile * TX;                         // first output column of this task
        const int d = chunk * 64 + lane;
        const bool valid = d < D;
        // plain build: one divergent region per task (only the last chunk is partial).  FUSE: every lane runs
        // (idle lanes shadow the last cell) because the wave reductions need the full wave; their stores are masked.
        if (FUSE || FINE || (SOFT && prob) || valid) {   // (wave reductions need the whole wave: idle lanes shadow the last cell, masked)
            const int dc = valid ? d : D - 1;
            const int dy = dc / p.wWin, dx = dc - dy * p.wWin;
            const px_t *lp = lds + dy * p.pitch + dx + tile * TX;
            const unsigned dbytes = (unsigned)d * 4u;

            // frame-0 rows come through the scalar cache; row r+1 is requested right after row r's
            // squared differences are done, so its latency hides behind the box sums and stores, and no
            // SMEM load is in flight while LDS results are waited for (SMEM returns out of order, which
            // would force a full lgkmcnt(0) drain per LDS read).
            const long long a_base = (long long)(y0 + oy) * p.W + (xt + ox);
            float av[C][NE];
#pragma unroll
            for (int c = 0; c < C; ++c) uload<NE>((cfptr)(I0 + a_base + c * HW), av[c]);

            float ring[U][TX];     // K==7: pair sums P; else: horizontal sums H
            float hprev[TX];
#pragma unroll
            for (int x = 0; x < TX; ++x) {
                hprev[x] = 0.f;
#pragma unroll
                for (int i = 0; i < U; ++i) ring[i][x] = 0.f;
            }

            for (int q = 0; q < NQ; ++q) {
#pragma unroll
                for (int m = 0; m < U; ++m) {
                    const int r = q * U + m;
                    const px_t *lr = lp + r * p.pitch;
                    float e[NE];
                    // LDS reads in NB batches: each batch is issued whole before any of it is consumed
                    constexpr int NB = (NE > 8) ? 2 : 1;
                    constexpr int BS = (NE + NB - 1) / NB;
#pragma unroll
                    for (int bb = 0; bb < NB; ++bb) {
                        px_t b[BS];
#pragma unroll
                        for (int s = 0; s < BS; ++s)
                            if (bb * BS + s < NE) b[s] = lr[bb * BS + s];
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (C == 3) {
                            // a (free) use of .w keeps each read a 16-B ds_read_b128 instead of the slower b96
#pragma unroll
                            for (int s = 0; s < BS; ++s)
                                if (bb * BS + s < NE) asm volatile("" ::"v"(b[s].w));
                        }
#pragma unroll
                        for (int s = 0; s < BS; ++s) {
                            if (bb * BS + s < NE) {
                                float a3[C];
#pragma unroll
                                for (int c = 0; c < C; ++c) a3[c] = av[c][bb * BS + s];
                                e[bb * BS + s] = sqdiff<C>(a3, b[s]);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    {   // next row's frame-0 values (row index clamped to the frame: the row after the
                        // last one is requested but never used)
                        const int rn = min(y0 + oy + r + 1, p.H - 1) - (y0 + oy);
                        cfptr an = (cfptr)(I0 + a_base + (long long)rn * p.W);
                        {
#pragma unroll
                            for (int c = 0; c < C; ++c) uload<NE>(an + c * HW, av[c]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    float h[TX];
                    hsum<K, TX>(e, h);
                    const bool emit = (K == 7) ? (q > 0) : (q > 0 || m == K - 1);   // r >= K-1
                    const int y = y0 + r - (K - 1);
                    const bool store_row = emit && y >= y0n;   // wave-uniform
                    const char *orow = (const char *)(out + ((long long)y * p.Wo + xt) * D);
                    float vrow[TX];
                    if constexpr (K == 7) {
#pragma unroll
                        for (int x = 0; x < TX; ++x) {
                            vrow[x] = (ring[m][x] + ring[(m + 2) % 6][x]) + (ring[(m + 4) % 6][x] + h[x]);
                            ring[(m + 5) % 6][x] = hprev[x] + h[x];   // P_{r-1}; its slot held P_{r-7}, consumed last row
                            hprev[x] = h[x];
                        }
                    } else {
#pragma unroll
                        for (int x = 0; x < TX; ++x) ring[m][x] = h[x];
#pragma unroll
                        for (int x = 0; x < TX; ++x) {
                            float t = ring[(m + 1) % K][x];   // oldest row first
#pragma unroll
                            for (int i = 2; i <= K; ++i) t += ring[(m + i) % K][x];
                            vrow[x] = t;
                        }
                    }
                    if constexpr (FINE) {
                        if (store_row) fine_epilogue<TX, FINE16, MID>(vrow, lane, y, xt, p.Wo, *fine);
                    } else
                    if (SOFT && prob && store_row) {   // (wave-uniform)
                        const char *prow = (const char *)(prob + ((long long)y * p.Wo + xt) * D);
#pragma unroll
                        for (int x = 0; x < TX; ++x) {
                            const float c = vrow[x];
                            const float m = wave_max_f32(valid ? -c : -INFINITY);
                            const float e = valid ? dfe_exp_nonpos(-c - m) : 0.f;
                            const float sum = wave_sum_f32_ordered(e);
                            if (valid) store_uniform_base(prow + (long long)x * D * 4, dbytes, e * (1.0f / sum));
                        }
                    } else if (SOFT && TX == 8 && store_row && D == 64 && p.stage_off > 0 && p.f16) {
                        // fp16 pyramid volume: the task row is 8 pixels x 128 B = 1 KB of contiguous volume -- through the wave's
                        // transpose scratch a lane picks up 8 adjacent cells, converts them (cost * scale, round to nearest even:
                        // v_cvt_f16_f32, not the truncating pkrtz) and the row leaves as ONE dwordx4 store per lane
                        float *xp = reinterpret_cast<float *>(dfe_smem + p.stage_off) + wave * (TX * 64);
#pragma unroll
                        for (int x = 0; x < TX; ++x) xp[x * 64 + lane] = vrow[x];
                        const f4_t lo = *reinterpret_cast<const f4_t *>(xp + 8 * lane), hi = *reinterpret_cast<const f4_t *>(xp + 8 * lane + 4);
                        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                        union { h2_t h[4]; f4_t v; } u;
                        const float sc = p.scale;
                        u.h[0] = h2_t{(_Float16)(lo[0] * sc), (_Float16)(lo[1] * sc)};
                        u.h[1] = h2_t{(_Float16)(lo[2] * sc), (_Float16)(lo[3] * sc)};
                        u.h[2] = h2_t{(_Float16)(hi[0] * sc), (_Float16)(hi[1] * sc)};
                        u.h[3] = h2_t{(_Float16)(hi[2] * sc), (_Float16)(hi[3] * sc)};
                        const char *orow_h = (const char *)out + ((long long)y * p.Wo + xt) * (D * 2);
                        if (p.stage_len)
                            asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"((unsigned)lane * 16u), "v"(u.v), "s"(orow_h) : "memory");
                        else
                            asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"((unsigned)lane * 16u), "v"(u.v), "s"(orow_h) : "memory");
                    } else if (!FUSE && TX == 8 && store_row && D == 64 && p.stage_off > 0) {
                        // one-chunk windows (the pyramid's 8 x 8): the task row is 8 pixels x 256 B = 2 KB of contiguous volume.
                        // Through a 2-KB LDS scratch of this wave it leaves as two dwordx4 stores (1 KB each) instead of eight
                        // dword stores of 256 B: a wave's stores issue one behind the other, and the launch is bound by that.
                        float *xp = reinterpret_cast<float *>(dfe_smem + p.stage_off) + wave * (TX * 64);
#pragma unroll
                        for (int x = 0; x < TX; ++x) xp[x * 64 + lane] = vrow[x];
                        const f4_t lo = *reinterpret_cast<const f4_t *>(xp + 4 * lane), hi = *reinterpret_cast<const f4_t *>(xp + 256 + 4 * lane);
                        // (non-temporal when the call's volumes exceed what the caches hold -- 1080p, 4 scales: 0.570 -> 0.471 ms per
                        //  pair; at VGA, where the cascade kernels find the 110 MB still cached, the hint costs 6 %)
                        if (p.stage_len)
                            asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\tglobal_store_dwordx4 %0, %3, %2 offset:1024 nt" ::"v"((unsigned)lane * 16u), "v"(lo), "s"(orow), "v"(hi) : "memory");
                        else
                            asm volatile("global_store_dwordx4 %0, %1, %2\n\tglobal_store_dwordx4 %0, %3, %2 offset:1024" ::"v"((unsigned)lane * 16u), "v"(lo), "s"(orow), "v"(hi) : "memory");
                    } else if (store_row) {
#pragma unroll
                        for (int x = 0; x < TX; ++x)
                            if (!FUSE || valid) store_uniform_base(orow + (long long)x * D * 4, dbytes, vrow[x]);
                        if constexpr (FUSE)
                            fuse_epilogue<TX>(vrow, valid, lane, chunk, (long long)(fa.row_off + y) * p.Wo + xt, fa);
                    }
                }
            }
        }
    }
}

template <int C, int K, int TX, int NT, int NW, int NQ, bool FUSE>
__global__ __launch_bounds__(NW * 64) void ssd_cv_tiled_kernel(const float *__restrict__ I0, const float *__restrict__ I1,
                                                               float *__restrict__ out, CvTiledArgs p, CvFuseArgs fa) {
    ssd_cv_tiled_body<C, K, TX, NT, NW, NQ, FUSE>(I0, I1, out, p, fa, blockIdx.x, blockIdx.y);
}

// Several independent frame pairs in one launch (the pyramid scales of the multiscale matcher): blockIdx.z = pair.  The
// coarser scales' grids are tiny (VGA scale 4: 30 blocks); launched one after the other each holds the GPU for ~13 us.
struct CvTiledMulti {
    const float *I0[DFE_MAX_RATIOS], *I1[DFE_MAX_RATIOS];
    float *out[DFE_MAX_RATIOS];
    float *prob[DFE_MAX_RATIOS];   // non-null: this pair leaves as soft-min probabilities (out[] is then not written)
    CvTiledArgs p[DFE_MAX_RATIOS];
    int gx[DFE_MAX_RATIOS], gy[DFE_MAX_RATIOS];
};
template <int C, int K, int TX, int NT, int NW, int NQ>
__global__ __launch_bounds__(NW * 64) void ssd_cv_tiled_multi_kernel(CvTiledMulti m) {
    // (last pair first: blocks are dispatched in z order, and the coarser scales' few blocks, which carry the soft-min
    //  epilogue and run ~3x longer, must not form the tail of the launch)
    const int z = gridDim.z - 1 - blockIdx.z;
    if ((int)blockIdx.x >= m.gx[z] || (int)blockIdx.y >= m.gy[z]) return;   // block-uniform
    ssd_cv_tiled_body<C, K, TX, NT, NW, NQ, false, true>(m.I0[z], m.I1[z], m.out[z], m.p[z], CvFuseArgs{}, blockIdx.x, blockIdx.y, m.prob[z]);
}

template <int C, int K, int TX, int NT, int NW, int NQ, bool F16, bool MID>
__global__ __launch_bounds__(NW * 64) void ssd_cv_tiled_fine_kernel(const float *__restrict__ I0, const float *__restrict__ I1, CvTiledArgs p, CvFineArgs fine) {
    ssd_cv_tiled_body<C, K, TX, NT, NW, NQ, false, false, true, F16, MID>(I0, I1, nullptr, p, CvFuseArgs{}, blockIdx.x, blockIdx.y, nullptr, &fine);
}

// geometry of one tiled launch
struct CvTilePlan {
    int TY, GX, lrows, lcols, pitch, nblocks, bpc;
    size_t lds_bytes;
    double score;   // fraction of the chip-time doing useful output rows (higher is better), 0 = not launchable
};

template <int C, int K, int TX, int NT, int NW>
static CvTilePlan plan_cv_tiled(int NQ, int Ho, int Wo, int hWin, int wWin, int ncu) {
    using px_t = typename Px<C>::type;
    constexpr int U = VUnroll<K>::value;
    CvTilePlan pl{};
    const int rows = U * NQ;
    pl.TY = rows - (K - 1);
    pl.GX = NT * TX;
    pl.lrows = rows + hWin - 1;
    pl.lcols = pl.GX + K - 1 + wWin - 1;
    const int M = Px<C>::bank_mod;
    pl.pitch = pl.lcols;                             // smallest pitch >= lcols with pitch == wWin (mod M):
    while ((pl.pitch - wWin) % M != 0) ++pl.pitch;   // keeps a wave that spans several dy rows conflict-free
    pl.lds_bytes = (size_t)pl.lrows * pl.pitch * sizeof(px_t);
    pl.score = 0;
    if (pl.TY < 1 || Ho < pl.TY || Wo < pl.GX || pl.lds_bytes > 160 * 1024) return pl;
    pl.nblocks = dfe_cdiv(Wo, pl.GX) * dfe_cdiv(Ho, pl.TY);
    int bpc = (int)((160 * 1024) / pl.lds_bytes);          // resident blocks per CU: LDS ...
    int by_waves = 4 / ((NW + 3) / 4);                      // ... and <= 128 VGPRs -> 4 waves/SIMD, ceil(NW/4) per block
    if (bpc > by_waves) bpc = by_waves;
    if (bpc < 1) bpc = 1;
    pl.bpc = bpc;
    long long slots = (long long)ncu * bpc;
    long long rounds = (pl.nblocks + slots - 1) / slots;
    double fill = (double)pl.nblocks / (double)(rounds * slots);       // tail quantisation
    // The kernel is bound by its store stream, not by its arithmetic (measured: halving the useful rows
    // per swept row does not slow it down), so row reuse only enters as a tie-breaker.
    double halo = 0.9 + 0.1 * (double)pl.TY / (double)rows;
    int D = hWin * wWin, nch = (D + 63) / 64, ntasks = NT * nch;
    double deal = (double)ntasks / (double)(((ntasks + NW - 1) / NW) * NW);   // idle waves in the last task round
    double occ = bpc * NW >= 16 ? 1.0 : (bpc * NW) / 16.0;             // too few waves cannot hide latency
    pl.score = fill * halo * deal * occ;
    return pl;
}

template <int C, int K, int TX, int NT, int NW, int NQ, bool FUSE = false>
static int launch_cv_tiled_one(dfe_ctx *ctx, const CvTilePlan &pl, const float *I0, const float *I1, int H, int W,
                               long long plane, int hWin, int wWin, float *out, const CvFuseArgs *fa = nullptr) {
    const int Ho = H - K + 1 - hWin + 1, Wo = W - K + 1 - wWin + 1;
    CvTiledArgs a;
    a.plane = plane;
    a.H = H; a.W = W; a.hWin = hWin; a.wWin = wWin; a.Ho = Ho; a.Wo = Wo;
    a.lrows = pl.lrows; a.lcols = pl.lcols; a.pitch = pl.pitch; a.seg_rows = 0; a.tile0_off = 0; a.stage_off = 0; a.stage_len = 0;
    a.chunk0 = ctx->cv_chunk0;
    auto kern = ssd_cv_tiled_kernel<C, K, TX, NT, NW, NQ, FUSE>;
    DFE_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes));
    dim3 grid(dfe_cdiv(Wo, pl.GX), dfe_cdiv(Ho, pl.TY));
    CvFuseArgs f{};
    if (fa) f = *fa;
    {
        DfeProfScope prof(ctx);
        hipLaunchKernelGGL(kern, grid, dim3(NW * 64), pl.lds_bytes, ctx->stream, I0, I1, out, a, f);
    }
    DFE_LAUNCH_CHECK(ctx);
    ctx->last_kernel = FUSE ? "ssd_cv_tiled_kernel+fused_tail" : "ssd_cv_tiled_kernel";
    return DFE_OK;
}

// picks the block shape -- (NT=4 tiles, 8 waves) or (NT=1, 6 waves), NQ in 2..5 row groups -- that wastes
// the least chip-time for this frame, then launches it
template <int C, int K, int TX>
static int launch_cv_tiled(dfe_ctx *ctx, const float *I0, const float *I1, int H, int W, long long plane, int hWin,
                           int wWin, float *out, bool *handled) {
    const int Ho = H - K + 1 - hWin + 1, Wo = W - K + 1 - wWin + 1;
    *handled = false;
    int best = 0;
    CvTilePlan bp{};
    for (int nq = 2; nq <= 5; ++nq) {
        if (ctx->cv_tyq >= 2 && ctx->cv_tyq <= 5 && nq != ctx->cv_tyq) continue;
        CvTilePlan p4 = plan_cv_tiled<C, K, TX, 4, 8>(nq, Ho, Wo, hWin, wWin, ctx->ncu);
        if (p4.score > bp.score) { bp = p4; best = 40 + nq; }
        CvTilePlan p1 = plan_cv_tiled<C, K, TX, 1, 6>(nq, Ho, Wo, hWin, wWin, ctx->ncu);
        if (p1.score > bp.score) { bp = p1; best = 10 + nq; }
        if constexpr (C == 3 && K == 7) {   // windows of one chunk (the pyramid's 8 x 8): 4 tasks on 4 waves, 4 blocks per CU
            if (hWin * wWin <= 64) {
                CvTilePlan p44 = plan_cv_tiled<C, K, TX, 4, 4>(nq, Ho, Wo, hWin, wWin, ctx->ncu);
                if (p44.score > bp.score) { bp = p44; best = 440 + nq; }
            }
        }
    }
    if (!best) return DFE_OK;   // no tile fits: caller falls back
    *handled = true;
    switch (best) {
        case 42: return launch_cv_tiled_one<C, K, TX, 4, 8, 2>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out);
        case 43: return launch_cv_tiled_one<C, K, TX, 4, 8, 3>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out);
        case 44: return launch_cv_tiled_one<C, K, TX, 4, 8, 4>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out);
        case 45: return launch_cv_tiled_one<C, K, TX, 4, 8, 5>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out);
        case 442: if constexpr (C == 3 && K == 7) return launch_cv_tiled_one<C, K, TX, 4, 4, 2>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out);
        case 443: if constexpr (C == 3 && K == 7) return launch_cv_tiled_one<C, K, TX, 4, 4, 3>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out);
        case 444: if constexpr (C == 3 && K == 7) return launch_cv_tiled_one<C, K, TX, 4, 4, 4>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out);
        case 445: if constexpr (C == 3 && K == 7) return launch_cv_tiled_one<C, K, TX, 4, 4, 5>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out);
        case 12: return launch_cv_tiled_one<C, K, TX, 1, 6, 2>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out);
        case 13: return launch_cv_tiled_one<C, K, TX, 1, 6, 3>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out);
        case 14: return launch_cv_tiled_one<C, K, TX, 1, 6, 4>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out);
        default: return launch_cv_tiled_one<C, K, TX, 1, 6, 5>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out);
    }
}

// fused build: the (4 tiles, 8 waves) block shape only; the caller falls back to build + tail pass otherwise
template <int C, int K, int TX>
static int launch_cv_tiled_fused(dfe_ctx *ctx, const float *I0, const float *I1, int H, int W, long long plane, int hWin,
                                 int wWin, float *out, const CvFuseArgs &fa, bool *handled) {
    const int Ho = H - K + 1 - hWin + 1, Wo = W - K + 1 - wWin + 1;
    *handled = false;
    int best = 0;
    CvTilePlan bp{};
    for (int nq = 2; nq <= 5; ++nq) {
        if (ctx->cv_tyq >= 2 && ctx->cv_tyq <= 5 && nq != ctx->cv_tyq) continue;
        CvTilePlan p4 = plan_cv_tiled<C, K, TX, 4, 8>(nq, Ho, Wo, hWin, wWin, ctx->ncu);
        if (p4.score > bp.score) { bp = p4; best = nq; }
    }
    if (!best) return DFE_OK;
    *handled = true;
    switch (best) {
        case 2: return launch_cv_tiled_one<C, K, TX, 4, 8, 2, true>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out, &fa);
        case 3: return launch_cv_tiled_one<C, K, TX, 4, 8, 3, true>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out, &fa);
        case 4: return launch_cv_tiled_one<C, K, TX, 4, 8, 4, true>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out, &fa);
        default: return launch_cv_tiled_one<C, K, TX, 4, 8, 5, true>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out, &fa);
    }
}


// ------------------------------------------------------------------------------------------
// row-image kernel: every cell of a tile row leaves through an LDS image as whole cache lines
// ------------------------------------------------------------------------------------------
// The tiled kernel above hides its arithmetic completely (160 us of compute at VGA) but is bound by its store
// pattern: each wave's 256-B piece has two partial cache lines that a sibling completes later, and partial-line
// writes are what the memory side is slow at (2.9-3.3 TB/s against 5.4 TB/s for line-aligned
// bursts; a partial line that misses L2 costs a read-modify-write at the ECC-protected HBM).
// In the reference layout the TX pixels of a tile row are ONE contiguous run of TX*D floats.  Here one block of 16
// waves owns that run completely and deposits it into a double-buffered LDS image whose float index is congruent to
// the global float index mod 32; after one LDS-only barrier per row the block copies the image out as 128-B-aligned
// dwordx4 bursts while the next row is already being computed.  Only the run's first and last line are partial.
//   * wave w sweeps chunk w (cells 64w..64w+63) exactly like a tiled-kernel task;
//   * cells 1024..1087 (a 17th chunk that has no wave) are swept as four 2-column quarter tasks by waves 0..3 -- one
//     per SIMD, so the extra work (0.46 of a task each) stays balanced -- with 12 extra registers of box-filter state;
//   * cells >= 1088 (one at 33x33, at most 8) are a lane-per-(cell, column) mini task of wave 4 (the first without a
//     quarter task).
// Frame-0 values: the tiled kernel's scalar loads go through L2, and behind this kernel's store stream their latency
// (waited for before every barrier) serialised compute with the copy-out (345 us -> 240 us without them).  Here the
// frame-0 tile sits in LDS as well; once per row each lane reads the pixel of column (lane & 15), and the subtract
// takes its frame-0 operand through DPP row_newbcast:s -- position s of the 14-wide window, no extra instruction,
// no SGPRs, no SMEM in the loop.
// Blocks are renumbered so that x-adjacent tiles -- which share the run's boundary lines -- run on the same XCD.
template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int S> __device__ __forceinline__ float bcast16(float v) {   // lane S of my row of 16, folded into the consumer
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + S, 0xf, 0xf, true));
}
template <int C, int S> __device__ __forceinline__ float sqdiff_bc(const typename Px<C>::type &a, const typename Px<C>::type &b) {
    if constexpr (C == 1) {
        float d = bcast16<S>(a) - b;
        return d * d;
    } else {
        float d0 = bcast16<S>(a.x) - b.x, d1 = bcast16<S>(a.y) - b.y, d2 = bcast16<S>(a.z) - b.z;
        return fmaf(d2, d2, fmaf(d1, d1, d0 * d0));
    }
}
// One row of a box-filter task over TXL columns: e[s] for the TXL+K-1 positions (frame-1 pixels lr[s] against frame-0
// positions s of `a`), horizontal K-sums, vertical K-sum through `ring` (phase M of the U-row cycle) -> v.
template <int C, int K, int TXL, int M, bool SM>
__device__ __forceinline__ void rowimg_task_row(const typename Px<C>::type *lr, const typename Px<C>::type &a,
                                                const float (&av)[C][TXL + K - 1], float (&ring)[VUnroll<K>::value][TXL],
                                                float (&v)[TXL]) {
    using px_t = typename Px<C>::type;
    constexpr int NEL = TXL + K - 1;
    constexpr int NB = (NEL > 10) ? 3 : (NEL > 5) ? 2 : 1;   // read batches: registers are the scarce resource here
    constexpr int BS = (NEL + NB - 1) / NB;
    float e[NEL];
    static_for<0, NB>([&](auto bbc) {
        constexpr int bb = decltype(bbc)::value;
        px_t b[BS];
#pragma unroll
        for (int s = 0; s < BS; ++s)
            if (bb * BS + s < NEL) b[s] = lr[bb * BS + s];
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (C == 3) {
#pragma unroll
            for (int s = 0; s < BS; ++s)
                if (bb * BS + s < NEL) asm volatile("" ::"v"(b[s].w));   // keep ds_read_b128
        }
        static_for<0, BS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (bb * BS + s < NEL) {
                if constexpr (SM) {
                    float a3[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) a3[c] = av[c][bb * BS + s];
                    e[bb * BS + s] = sqdiff<C>(a3, b[s]);
                } else {
                    e[bb * BS + s] = sqdiff_bc<C, bb * BS + s>(a, b[s]);
                }
            }
        });
        __builtin_amdgcn_sched_barrier(0);
    });
    float h[TXL];
    hsum_vh<K, TXL>(e, h);
    if constexpr (K == 7) {
        // ring holds the pair sums P_{r-5}..P_{r-1} and, in the slot whose P has just been consumed, H_{r-1}
        // (which becomes P_{r-1} = H_{r-1} + H_r in place)
#pragma unroll
        for (int x = 0; x < TXL; ++x) {
            v[x] = (ring[M][x] + ring[(M + 2) % 6][x]) + (ring[(M + 4) % 6][x] + h[x]);
            ring[(M + 5) % 6][x] += h[x];
            ring[M][x] = h[x];
        }
    } else {
#pragma unroll
        for (int x = 0; x < TXL; ++x) ring[M][x] = h[x];
#pragma unroll
        for (int x = 0; x < TXL; ++x) {
            float t = ring[(M + 1) % K][x];   // oldest row first
#pragma unroll
            for (int i = 2; i <= K; ++i) t += ring[(M + i) % K][x];
            v[x] = t;
        }
    }
}

// columns of the frame-0 tile in LDS: the NE = 14 positions of a task row, read at up to +6 (quarter tasks) + 15 (the
// row of 16 a DPP broadcast reaches)
constexpr int kT0W = 24;
// compile-time LDS geometry of the 33 x 33 instantiations (what rowimg_plan / launch_cv_rowimg_sweep compute at run time)
// the lane id from the execution mask (v_mbcnt): two instructions and no live register -- the row loop's branches re-derive
// their lane geometry from this instead of holding threadIdx.x (or spilling it: every register counts in the 3-channel sweeps)
__device__ __forceinline__ int lane_id_fresh() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
template <int C, int K, int TX> struct RowimgGeom {
    static constexpr int R = 64, R0 = 8;                                   // column sweep: rows of the frame-1 / frame-0 rings
    static constexpr int lcols33 = TX + K - 1 + 32;
    static constexpr int M = Px<C>::bank_mod;
    static constexpr int pitch33 = 33 + (lcols33 - 33 + M - 1) / M * M;    // smallest pitch >= lcols with pitch == wWin (mod M)
    static constexpr int px_bytes = sizeof(typename Px<C>::type);
    static constexpr int sweep_tile0_off = (R * pitch33 * px_bytes + 127) / 128 * 128;
    static constexpr int sweep_stage_off = sweep_tile0_off + (R0 * kT0W * px_bytes + VUnroll<K>::value * 64 * 4 + 127) / 128 * 128;
    static constexpr int stage_len33 = (TX * 1089 + 32 + 31) / 32 * 32;
    static constexpr int stage_len33h = (TX * 1089 + 64 + 31) / 32 * 32;   // fp16 volume: image congruent to the global index mod 64 (a line of halfs)
};
// DC: the window's cell count as a compile-time constant (0 = run time).  With D known, the 8 deposit addresses per task row
// (st + x*D + d), the scan's and the copy-out's become immediate offsets -- instructions of the lock-stepped phases.
// F16: the volume leaves as fp16 (cost * p.scale, round to nearest even).  The row image stays fp32 -- the fused arg-min scan
// reads it, so indices and minima are those of the fp32 build ("arg-min before the down-convert") -- and is kept congruent to
// the global HALF index mod 64; the copy-out converts 8 cells per lane into one dwordx4.  Static tiles only (a 128-B line
// holds 64 cells: completing the run's last line would take up to 63 cells of the next pixel, more than the mini task's
// idle lanes, so partial lines stay and rely on the XCD-aware block order).
template <int C, int K, int TX, bool SM, bool FUSE, bool SWEEP, int DC = 0, bool F16 = false>
__global__ __launch_bounds__(1024) void ssd_cv_rowimg_kernel(const float *__restrict__ I0, const float *__restrict__ I1,
                                                             float *__restrict__ out, CvTiledArgs p, CvFuseArgs fa) {
    using px_t = typename Px<C>::type;
    constexpr int NW = 16;
    constexpr int U = VUnroll<K>::value;
    constexpr int R0 = 8;                        // column sweep: rows of the frame-0 ring
    constexpr int LW = NW - 1;                   // column sweep: the wave that streams the tiles and never stores
    constexpr int NE = TX + K - 1;
    constexpr int NQW = DFE_NQW;                 // waves that carry a share of the 17th chunk (tuning: 4 or 8)
    constexpr int TQ = TX / NQW;                 // columns of such a share ("quarter task")
    // Fused column sweep: the waves split the work behind the barrier -- waves 0..7 scan one pixel's run each for its minimum,
    // waves 8..14 copy the image out (wave 15 refills the rings) -- instead of every wave doing a half-pixel scan AND three
    // pieces of the copy-out before it can start on the next row: per wave the chain scan -> stores (each wave's own stores
    // issue one behind the other, 130+ cycles apiece, more when the memory pushes back) -> main task was the row's critical
    // path, 2.7 of 5.1 kilocycles of a VGA row spent before the slowest waves began their main task (s_memtime stamps of two blocks, round 2).
    // (3-channel frames only: with one channel the main task is half as long and the all-waves scan of the plain layout
    //  wins, VGA luminance 215 against 254 us)
    constexpr bool ROLES = FUSE && (SWEEP || DFE_ROLES_STATIC) && DFE_ROLES && TX == 8 && DC == 1089 && C == 3;
    // the last wave neither scans nor copies: in the column sweep it refills the rings, with roles it carries the mini task
    constexpr bool HAS_XW = SWEEP || ROLES;
    // (Which SIMD carries what, wave w on SIMD w % 4: every SIMD has one quarter-task + scan wave (0 .. 3) and one plain scan wave (4 .. 7);
    //  SIMD 3 also has wave 15 with the mini task and the ring refill, ~93 vector instructions a row on top of its main task
    //  (tools/isa_regions.py).  Round 4 tried pixel 7's scan on wave 8 (SIMD 0) with wave 7 copying instead: no difference, r04_d.)
    constexpr int QW0 = 0;                       // first wave with a quarter task (on the copy waves, 8..11, instead: 265 against 258 us)
    static_assert(NE <= 16, "row_newbcast reaches 16 positions");
    static_assert(TX % NQW == 0, "whole columns per quarter task");
    // 33 x 33 instantiation: the LDS geometry is a compile-time constant (rowimg_plan / launch_cv_rowimg_sweep compute the same
    // numbers) -- fewer scalars to keep, immediate offsets instead of address arithmetic
    constexpr bool CG = DC == 1089;
    const int g_wWin = CG ? 33 : p.wWin, g_hWin = CG ? 33 : p.hWin;
    const int g_lcols = CG ? TX + K - 1 + 32 : p.lcols;
    const int g_pitch = CG ? RowimgGeom<C, K, TX>::pitch33 : p.pitch;
    const int g_lrows = (CG && SWEEP) ? RowimgGeom<C, K, TX>::R : p.lrows;
    const int g_tile0_off = (CG && SWEEP) ? RowimgGeom<C, K, TX>::sweep_tile0_off : p.tile0_off;
    const int g_stage_off = (CG && SWEEP) ? RowimgGeom<C, K, TX>::sweep_stage_off : p.stage_off;
    const int g_stage_len = CG ? (F16 ? RowimgGeom<C, K, TX>::stage_len33h : RowimgGeom<C, K, TX>::stage_len33) : p.stage_len;
    static_assert(!(F16 && SWEEP), "fp16 volumes: static tiles only");
    constexpr int LM = F16 ? 63 : 31;                    // cells per 128-B line of the volume, minus one
    px_t *lds = reinterpret_cast<px_t *>(dfe_smem);                             // frame-1 tile [lrows][pitch]
    const px_t *t0 = reinterpret_cast<const px_t *>(dfe_smem + g_tile0_off);    // frame-0 tile [ROWS][32]
    float *stage = reinterpret_cast<float *>(dfe_smem + g_stage_off);          // [2][stage_len], 128-B aligned

    const long long HW = p.plane;
    const int oy = (g_hWin - 1) >> 1, ox = (g_wWin - 1) >> 1;
    // Tail-line ownership (33 x 33 instantiation): the run of 8 pixels ends inside a 128-B line whose other part belongs to the
    // NEXT pixel of the image row.  Two blocks writing the two parts of such a line only works while both parts meet in one
    // L2 (partial lines that miss L2 cost a read-modify-write at the memory, section 4.2), which ties the schedule to the
    // XCD layout.  Instead the block also computes the <= 31 leading cells of pixel x0+8 that complete its last line (31
    // idle lanes of the mini task: one output per lane, no extra instructions) and writes that line whole; its right
    // neighbour skips its head.  Left: two partial lines per IMAGE row (first and last column) instead of two per tile row.
    constexpr bool TOWN = DC == 1089 && !F16;
    // Static tiles: one piece per block, XCD-aware block order (hardware deals linear block ids round-robin to the 8 XCDs; give
    // every XCD a contiguous range of tiles, x fastest, so neighbours in x share an L2), TY output rows per block, the last
    // tile row shifted inwards.  Column sweep: the block walks down its range of the column-major (column, row) sequence
    // (sweep_cut), one piece per column it touches; the frame-1 tile is a ring of g_lrows (= 64) rows that wave LW keeps
    // filled a row step ahead, so the K-1 warm-up rows are paid once per piece instead of once per TY rows.
    // (static tiles: p.seg_rows is the tile height, a run-time value -- one instantiation serves every height)
    int pos = 0, pend = 1, bx = 0, by = 0;
    if constexpr (SWEEP) {
        const int ncols = (p.Wo + TX - 1) / TX;
        pos = sweep_cut(blockIdx.x, gridDim.x, ncols, p.Ho, p.sw_ovh, p.sw_min);
        pend = sweep_cut(blockIdx.x + 1, gridDim.x, ncols, p.Ho, p.sw_ovh, p.sw_min);
        if (pos >= pend) return;                                     // block-uniform
    } else {
        const int nb = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
        const int per = nb >> 3, rem = nb & 7, xcd = lin & 7, slot = lin >> 3;
        const int t = xcd * per + min(xcd, rem) + slot;      // XCDs 0..rem-1 own per+1 tiles, the others per
        by = t / (int)gridDim.x;
        bx = t - by * (int)gridDim.x;
    }
  for (;;) {   // pieces of this block (static tiles: one)
    // Every per-lane quantity is derived afresh from the thread id in each piece: nothing per-lane is then live across the
    // piece loop's back edge, and the register allocation of a piece is that of a one-piece kernel (with the ids taken
    // outside the loop the hoisted lane geometry pushed the 3-channel sweep 50 registers into scratch).
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int prows = p.seg_rows;
    if constexpr (SWEEP) {
        bx = pos / p.Ho;
        by = pos - bx * p.Ho;                                        // first output row of the piece
        prows = min(pend - pos, p.Ho - by);
    }
    const int x0n = bx * TX, y0n = SWEEP ? by : by * p.seg_rows;
    const int x0 = min(x0n, p.Wo - TX), y0 = SWEEP ? y0n : min(y0n, p.Ho - p.seg_rows);
    const int nsweep = min(prows, p.Ho - y0) + (K - 1);        // rows this piece sweeps
    const int t0rows = SWEEP ? R0 : nsweep;                    // rows of the frame-0 tile
    const int nover = x0n - x0;                                // leading columns of a shifted last tile: its neighbour's, not stored from here
    const bool has_next = TOWN && x0 + TX < p.Wo;              // the pixel after the run is in the same image row: own the tail line
    const bool skip_head = TOWN && x0n > 0;                    // ... and the left neighbour owns the line my (stored) run starts in

    const int ls = lane;
    for (int r = wave; r < g_lrows; r += NW) {
        const float *src = I1 + (long long)min(y0 + r, p.H - 1) * p.W + x0;
        for (int s = ls; s < g_lcols; s += 64) {
            if constexpr (C == 1) {
                lds[r * g_pitch + s] = src[s];
            } else {
                lds[r * g_pitch + s] = make_float4(src[s], src[HW + s], src[2 * HW + s], 0.f);
            }
        }
    }
    {   // frame-0 tile: rows y0+oy.., columns x0+ox..x0+ox+NE (one more than a task row's NE: the tail cells belong to
        // pixel x0+TX), padded to kT0W columns (quarter tasks read at +2w)
        px_t *t0w = reinterpret_cast<px_t *>(dfe_smem + g_tile0_off);
        for (int r = wave; r < t0rows; r += NW) {
            if (ls < kT0W) {
                const float *src = I0 + (long long)min(y0 + oy + r, p.H - 1) * p.W + (x0 + ox) + min(ls, NE);
                if constexpr (C == 1) {
                    t0w[r * kT0W + ls] = src[0];
                } else {
                    t0w[r * kT0W + ls] = make_float4(src[0], src[HW], src[2 * HW], 0.f);
                }
            }
        }
    }
    const long long a_base = (long long)(y0 + oy) * p.W + (x0 + ox);
    __syncthreads();

    const int D = DC ? DC : g_hWin * g_wWin;
    const int RUN = TX * D;                       // floats in the block's run
    const int l16 = lane & 15;
    float av[C][NE];
    if constexpr (SM && !DFE_SMEM_JIT) {
#pragma unroll
        for (int c = 0; c < C; ++c) uload<NE>((cfptr)(I0 + a_base + c * HW), av[c]);
    }

    // main task: chunk `wave`
    // (lane predicates that do not change from row to row are either compile-time constants -- D = 1089: every lane of a
    //  chunk and of the quarter tasks is inside the window -- or recomputed at their use from a laundered lane id: hoisted
    //  out of the row loop each of them holds an SGPR pair for the whole sweep, and the scalar file is what this kernel
    //  runs out of: 42 frame-0 scalars + pointers; a spilled pair costs two v_readlane per use)
    const int d = wave * 64 + lane;
    const bool valid = DC >= 1024 ? true : d < D;
    // lane address of a task: static tiles lds + (dy + r)*pitch + dx; column sweep lds + ((dy + r) & (lrows-1))*pitch + dx,
    // kept as the pair (dy, dx + column offset) packed into one register
    int lp;
    {
        const int dc = valid ? d : D - 1;         // idle lanes shadow the last cell, their deposits are masked
        const int dy = dc / g_wWin, dx = dc - dy * g_wWin;
        lp = SWEEP ? (dy << 16 | dx) : dy * g_pitch + dx;
    }
    // Column sweep: a task's lane address is carried as a byte offset into the ring and stepped one ring row per sweep row
    // (add, subtract the ring size, unsigned minimum: three full-rate ops, no multiply, no per-row unpacking of (dy, dx)).
    const unsigned ringB = (unsigned)(g_lrows * g_pitch) * (unsigned)sizeof(px_t), pitchB = (unsigned)g_pitch * (unsigned)sizeof(px_t);
    auto ring_off0 = [&](int packed) -> int {   // (dy, dx) -> byte offset of ring row dy, column dx (sweep row 0)
        return (int)((unsigned)(packed >> 16) * pitchB + (unsigned)(packed & 0xffff) * (unsigned)sizeof(px_t));
    };
    auto ring_step = [&](int &off) {
        const unsigned t = (unsigned)off + pitchB;
        off = (int)min(t, t - ringB);           // t < ringB: t - ringB wraps to a huge number
    };
    auto row_ptr = [&](int packed, int r) -> const px_t * {
        if constexpr (SWEEP)
            return reinterpret_cast<const px_t *>(reinterpret_cast<const char *>(lds) + packed);
        else
            return lds + packed + r * g_pitch;
    };
    // quarter task (waves 0..3): cells 1024 + lane, columns TQ*wave .. TQ*wave + TQ-1
    const bool has_q = wave >= QW0 && wave < QW0 + NQW && D > 1024;    // wave-uniform
    const int dq = 1024 + lane;
    const bool validq = DC >= 1088 ? true : dq < D;
    int lpq;
    {
        const int dc = validq ? dq : D - 1;
        const int dy = dc / g_wWin, dx = dc - dy * g_wWin + TQ * (wave - QW0);
        lpq = SWEEP ? (dy << 16 | dx) : dy * g_pitch + dx;
    }
    // mini task (wave 4): cells 1088 + (lane >> 3), column lane & 7 -- one output per lane; with tail-line ownership its
    // lanes 32..62 are the cells 0..30 of the pixel after the run (column TX)
    // the mini task's wave: the first one behind the quarter tasks; 3-channel column sweeps: the wave that refills the rings
    // (it neither scans nor copies; with the mini task on wave 4, behind that wave's main task and scan, it was the last one
    //  at the barrier: VGA fused 258 -> 250 us, plain 228 -> 222 us)
    constexpr int MW = (C == 3 && (ROLES || (SWEEP && !FUSE))) ? LW : QW0 + NQW;
    const bool has_m = wave == MW && (D > 1088 || TOWN);    // wave-uniform
    const bool mtail = TOWN && lane >= 32;
    const int dm = mtail ? lane - 32 : 1088 + (lane >> 3), xm = mtail ? TX : lane & 7;
    const bool validm = mtail ? lane < 63 : (dm < D && xm < TX && (!TOWN || lane < 32));
    int lpm;
    {
        const int dc = validm ? dm : D - 1;
        const int dy = dc / g_wWin, dx = dc - dy * g_wWin + xm;
        lpm = SWEEP ? (dy << 16 | dx) : dy * g_pitch + dx;
    }

    if constexpr (SWEEP) { lp = ring_off0(lp); lpq = ring_off0(lpq); lpm = ring_off0(lpm); }
    // waves with an extra task get issue priority: they run ahead while their three SIMD-mates fill the gaps, instead of
    // finishing their surplus alone (one wave per SIMD hides no latency) while 15 waves sit at the barrier.
    // (Tried and dropped: letting the odd waves run a row's flow epilogue after the barrier, from the row image, to
    //  de-phase LDS reads and reductions across the lock-stepped waves -- 13 % slower.)
    if (has_q) __builtin_amdgcn_s_setprio(3);
    else if (has_m) __builtin_amdgcn_s_setprio(2);
    else if (SWEEP && DFE_LW_PRIO && wave == LW) __builtin_amdgcn_s_setprio(DFE_LW_PRIO);
    float ring[U][TX], ringq[U][TQ];
    // the mini task's box-filter state lives in LDS ([U][64] floats behind the frame-0 tile; only wave 4 touches it)
    float *rm = reinterpret_cast<float *>(dfe_smem + g_tile0_off + t0rows * kT0W * sizeof(px_t)) + lane;
    float hold[DFE_REFILL_AHEAD][C];              // column sweep, wave LW: the tile pixels requested DFE_REFILL_AHEAD row steps ago
#pragma unroll
    for (int i = 0; i < DFE_REFILL_AHEAD; ++i)
#pragma unroll
        for (int c = 0; c < C; ++c) hold[i][c] = 0.f;
#pragma unroll
    for (int i = 0; i < U; ++i) {
#pragma unroll
        for (int x = 0; x < TX; ++x) ring[i][x] = 0.f;
#pragma unroll
        for (int x = 0; x < TQ; ++x) ringq[i][x] = 0.f;
        if (wave == MW) rm[i * 64] = 0.f;
    }

    // Flow epilogue from the row image (FUSE): after the barrier of row r the run of every pixel of that row is complete in
    // image (r & 1), so two waves per pixel scan it -- every lane 9 consecutive cells (strict '<' keeps the lane's first
    // minimum), one wave minimum, the lowest lane attaining it owns the first index; half 0 / half 1 go to the two planes
    // of fa.part, finalize keeps the smaller, the first on ties.  ~55 VALU per wave and row, but a chain (LDS latency, min
    // tree, 6 DPP steps, ballot, readlane) that nothing overlaps where it sits, between the barrier and the copy-out:
    // 0.39 us of a 2.06 us row step (ablation, VGA, persistent sweep).  Moving it does not help: run for row r-1 inside the
    // sweep of row r on the twelve waves without a quarter task (one or two half-pixel units each, at the end of their row
    // body, the last row of a piece scanned after the loop) it was bit-identical and 9 % SLOWER (316 against 290 us) -- the
    // chain then sits on the critical path of 12 waves instead of being shared by 16; after the copy-out, split around
    // it, or with its reads issued with the copy-out's it measured +3..5 % (round 1).
    auto scan_row = [&](const float *stp, long long pgp) {   // stp: image of the row to scan (+ its a0), pgp: its first entry in the planes
        constexpr int CPL = 9;                                         // cells per lane: 128 lanes x 9 >= 1096
        int lsc = lane;
        asm volatile("" : "+v"(lsc));
        {
            const int xx = wave >> 1, hh = wave & 1;
            const int c0 = (hh * 64 + lsc) * CPL;
            const float *px = stp + xx * D;
            int cv[CPL];
            bool lane_out = false;                                     // fast path (D = 9 n): this lane lies entirely outside the window
            if (D % CPL == 0) {                                        // block-uniform
                lane_out = c0 >= D;
                const int *pc = reinterpret_cast<const int *>(px) + (lane_out ? 0 : c0);
#pragma unroll
                for (int i = 0; i < CPL; ++i) cv[i] = pc[i];           // all reads in flight
            } else {
#pragma unroll
                for (int i = 0; i < CPL; ++i) cv[i] = __float_as_int(px[min(c0 + i, D - 1)]);
#pragma unroll
                for (int i = 0; i < CPL; ++i) cv[i] = c0 + i < D ? cv[i] : 0x7f800000;
            }
            static_assert(CPL == 9, "min tree below is written for 9 cells");
            int best = min(min(min(cv[0], cv[1]), min(cv[2], cv[3])), min(min(min(cv[4], cv[5]), min(cv[6], cv[7])), cv[8]));
            int bi = CPL - 1;
#pragma unroll
            for (int i = CPL - 2; i >= 0; --i) bi = cv[i] == best ? i : bi;
            best = lane_out ? 0x7f800000 : best;
            const int vmin = wave_min1(best);
            const int f = __builtin_ctzll(__builtin_amdgcn_ballot_w64(best == vmin));
            const int bif = __builtin_amdgcn_readlane(bi, f);
            if (lsc == 0 && xx >= nover)
                fa.part[(long long)hh * fa.Ptot + pgp + xx] = make_float2(__int_as_float(vmin), __int_as_float((hh * 64 + f) * CPL + bif));
        }
    };
    // ROLES: one wave per pixel.  121 lanes' worth of 9 cells = lanes 0..63 (cells 9 l ..) and a second unit on lanes 0..56
    // (cells 9 (64 + l) ..); both units' reads in flight together, ONE wave minimum for both, the first index from the lower
    // unit if any of its lanes attains the minimum.  Plane 1 of fa.part gets +inf (finalize keeps the smaller of the two).
    float *recbuf = stage + 2 * g_stage_len;       // ROLES: [2][DFE_REC] the tile row's record, double-buffered like the images
    auto scan_row_whole = [&](const float *stp, int rpar) {
        constexpr int CPL = 9;
        static_assert(!ROLES || DC % CPL == 0, "whole lanes only");
        int lsc = lane;
        asm volatile("" : "+v"(lsc));
        const int xx = wave;
        const bool out1 = (64 + lsc) * CPL >= D;
        const int *pc0 = reinterpret_cast<const int *>(stp + xx * D) + lsc * CPL;
        const int *pc1 = pc0 + (out1 ? 0 : 64 * CPL);
        int c0[CPL], c1[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) c0[i] = pc0[i];
#pragma unroll
        for (int i = 0; i < CPL; ++i) c1[i] = pc1[i];
        int b0 = min(min(min(c0[0], c0[1]), min(c0[2], c0[3])), min(min(min(c0[4], c0[5]), min(c0[6], c0[7])), c0[8]));
        int b1 = min(min(min(c1[0], c1[1]), min(c1[2], c1[3])), min(min(min(c1[4], c1[5]), min(c1[6], c1[7])), c1[8]));
        b1 = out1 ? 0x7f800000 : b1;
        const int vmin = wave_min1(min(b0, b1));
        // the cell index is looked for only in the unit that holds the first minimum (wave-uniform choice): the lower unit if
        // any of its lanes attains it
        const unsigned long long m0 = __builtin_amdgcn_ballot_w64(b0 == vmin);
        int idx;
#if DFE_SCAN_SIDX
        // The first cell of the first lane that attains the minimum.  Only that lane's answer is wanted, so the search is scalar: one
        // compare per cell into a lane mask, bit f of it tested on the scalar unit -- 8 vector instructions instead of 8 compares + 8
        // selects + a v_readlane (round 4: the scan waves are as close to the row's critical path as the copy waves).
        if (m0) {
            const int f = __builtin_ctzll(m0);
            int i0 = CPL - 1;
#pragma unroll
            for (int i = CPL - 2; i >= 0; --i) i0 = ((__builtin_amdgcn_ballot_w64(c0[i] == vmin) >> f) & 1) ? i : i0;
            idx = f * CPL + i0;
        } else {
            const int f = __builtin_ctzll(__builtin_amdgcn_ballot_w64(b1 == vmin));
            int i1 = CPL - 1;
#pragma unroll
            for (int i = CPL - 2; i >= 0; --i) i1 = ((__builtin_amdgcn_ballot_w64(c1[i] == vmin) >> f) & 1) ? i : i1;
            idx = (64 + f) * CPL + i1;
        }
#else
        if (m0) {
            int i0 = CPL - 1;
#pragma unroll
            for (int i = CPL - 2; i >= 0; --i) i0 = c0[i] == vmin ? i : i0;
            const int f = __builtin_ctzll(m0);
            idx = f * CPL + __builtin_amdgcn_readlane(i0, f);
        } else {
            int i1 = CPL - 1;
#pragma unroll
            for (int i = CPL - 2; i >= 0; --i) i1 = c1[i] == vmin ? i : i1;
            const int f = __builtin_ctzll(__builtin_amdgcn_ballot_w64(b1 == vmin));
            idx = (64 + f) * CPL + __builtin_amdgcn_readlane(i1, f);
        }
#endif
        // The pixel's minimum, first index and centre cost go into the tile row's RECORD in LDS (8 x (min, idx) | 8 x centre | 0 ...);
        // behind the NEXT barrier one wave writes the 128-B record out whole (write_record below).  Until round 3 every scan wave
        // stored its pixel's entries into three planes itself: two 8-B entries, 64 B of lead cells, 4 B of centre per pixel and row
        // step -- partial lines that neighbouring blocks on other XCDs complete, and the reason the fused 1080p kernel took
        // 1.78 .. 2.21 ms depending on where the process' arena had landed (without them: 1.78 .. 1.81 ms in every process).
        // The pixel's first DFE_REC_NLEAD cells follow in the same record (lane 0 holds them already).
        static_assert(!ROLES || DC == 1089, "centre cell 544");
        {
            const float cen = stp[xx * D + 544];
            if (lsc == 0) {
                float *rb = recbuf + rpar * DFE_REC;
                rb[2 * xx] = __int_as_float(vmin);
                rb[2 * xx + 1] = __int_as_float(idx);
                rb[DFE_REC_CENTRE + xx] = cen;
                if constexpr (DFE_REC_NLEAD >= 4) {      // lane 0 holds the run's cells 0..8 already: the first 4 / 8 ride along (16-B LDS writes)
                    static_assert(DFE_REC_NLEAD == 4 || DFE_REC_NLEAD == 8, "one or two 16-B pieces");
                    typedef int i4_t __attribute__((ext_vector_type(4)));
                    i4_t *lp4 = reinterpret_cast<i4_t *>(rb + DFE_REC_LEAD + DFE_REC_NLEAD * xx);
                    lp4[0] = i4_t{c0[0], c0[1], c0[2], c0[3]};
                    if constexpr (DFE_REC_NLEAD == 8) lp4[1] = i4_t{c0[4], c0[5], c0[6], c0[7]};
                }
            }
        }
    };
    // the record of the tile row whose scan ran behind the previous barrier: [column group bx][output row yrow of the pair]
    // (rb: the record's address, carried from row to row like G0 -- one 64-bit scalar instead of the base, the row pitch and the band
    //  offset live across the sweep: this kernel has no scalar or vector register to spare, and a spill inside the row loop costs
    //  the 1080p launch, whose scratch lines do not survive in L2 behind the 8.6-GB stream, 0.6 ms)
    auto write_record = [&](int rpar, const float *rb) {
        int lw = lane;
        asm volatile("" : "+v"(lw));
        if (lw < DFE_REC / 4) {
            const f4_t v = *reinterpret_cast<const f4_t *>(recbuf + rpar * DFE_REC + 4 * lw);
            asm volatile("global_store_dwordx4 %0, %1, %2" DFE_REC_ST_FLAGS ::"v"((unsigned)lw * 16u), "v"(v), "s"(rb) : "memory");
        }
    };
    if constexpr (ROLES) {   // the record's tail (8 zeros per buffer) is written once per piece; the scans fill the rest
        if (wave == 0 && lane < 16) recbuf[(lane >> 3) * DFE_REC + 24 + (lane & 7)] = 0.f;
    }
    // (the run's global position is carried in BYTES: the image offset a0 is then one mask of its low word and the copy waves add it to
    //  the output pointer as it is -- round 5: every scalar instruction of the row loop costs ~0.2 % of the kernel)
    constexpr int ES = F16 ? 2 : 4;                                          // bytes per cell of the volume
    long long G0b_run = ((long long)(y0 - (K - 1)) * p.Wo + x0) * D * ES;    // row r = 0 is output row y0 - (K-1) (a warm-up row, not stored)
    const long long G0b_step = (long long)p.Wo * D * ES;
    // (b) rows from which on the piece stores: r >= K - 1 (the box filter is warm) and output row y0 + r - (K-1) >= y0n (a shifted last
    //  static tile leaves its first rows to its neighbour) -- one per-piece scalar instead of two compares and an and per row
    const int r_store = (K - 1) + max(0, y0n - y0);
    // (a) byte offset of the NEXT row's frame-0 window from a_base, carried: + W * 4 per row.  No clamp at the piece's end: the row after
    //  the last sweep row, y0 + oy + nsweep <= Ho + oy + K - 1 = H - ceil((hWin - 1) / 2), is inside the frame for every window of >= 2
    //  rows (its values are loaded and never used).
    unsigned a_roff = (unsigned)p.W * 4u;
    long long pg_next = FUSE ? (long long)(fa.row_off + y0 - (K - 1)) * p.Wo + x0 : 0;
    // ROLES: record of the row BEFORE row r = 0 of this piece (output row y0 - K of column group bx); + DFE_REC per row step
    const float *rec_prev = ROLES ? fa.rec + ((long long)bx * fa.rec_rows + (fa.row_off + y0 - (K - 1) - 1)) * DFE_REC : nullptr;
    const int nq = (nsweep + U - 1) / U;
    for (int q = 0; q < nq; ++q) {
        static_for<0, U>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const int r = q * U + m;
            if (SWEEP && r >= nsweep) return;                                // block-uniform (static tiles: nsweep is a multiple of U)
            const int y = y0 + r - (K - 1);
            const bool store_row = r >= r_store;                             // block-uniform
            // (G0 = ((long long)y * p.Wo + x0) * D, the global float index of the run, and pg_run, the row's first entry of the
            //  fused per-pixel planes, are carried from row to row: one 64-bit add each instead of two 64-bit multiplies)
            const long long G0b = G0b_run;
            G0b_run += G0b_step;     // (tried as a 32-bit unsigned step, one scalar fewer: the allocator answered with a vector spill reloaded in every row)
            const long long pg_run = pg_next;
            if constexpr (FUSE) pg_next += p.Wo;
            const int a0 = (int)(((unsigned)G0b / (unsigned)ES) & (unsigned)LM);
            float *st = stage + (r & 1) * g_stage_len + a0;                  // image of the run, congruent mod 32
            [[maybe_unused]] const int rn = min(r + 1, nsweep - 1);   // next frame-0 row of the non-A32 form (clamped: the row after the last is never used)
            const int t0r = (SWEEP ? (r & (R0 - 1)) : r) * kT0W;
            auto do_main = [&]() {
                {
                    px_t a{};
                    if constexpr (!SM) a = t0[t0r + l16];
                    float v[TX];
                    if constexpr (SM && DFE_SMEM_JIT) {   // this row's scalars, just in time: they land while the first LDS batch is in flight
#pragma unroll
                        for (int c = 0; c < C; ++c) uload<NE>((cfptr)(I0 + a_base + (long long)r * p.W + c * HW), av[c]);
                    }
                    rowimg_task_row<C, K, TX, m, SM>(row_ptr(lp, r), a, av, ring, v);
                    if constexpr (SWEEP) ring_step(lp);
                    if constexpr (SM && !DFE_SMEM_JIT) {   // next row's scalars (requested behind the last squared difference instead --
                                                           // inside the task row, in front of the sums -- the 42 scalars are live across the
                                                           // sums as well and the 3-channel sweeps spill: 62 / 99 scalars, 104 / 124 B of scratch)
#if DFE_A32
                        // (32-bit unsigned row / plane offsets next to ONE 64-bit base: the scalar loads take them as soffset, and the
                        //  sweep keeps 3 scalars fewer than with a running 64-bit pointer and two 64-bit plane offsets)
                        const char *ab = reinterpret_cast<const char *>(I0 + a_base);
                        const unsigned roff = a_roff, hwb = (unsigned)HW * 4u;
                        a_roff += (unsigned)p.W * 4u;
#pragma unroll
                        for (int c = 0; c < C; ++c) uload<NE>((cfptr)(ab + (unsigned long long)(roff + (unsigned)c * hwb)), av[c]);
#else
#pragma unroll
                        for (int c = 0; c < C; ++c) uload<NE>((cfptr)(I0 + a_base + (long long)rn * p.W + c * HW), av[c]);
#endif
                    }
                    // deposit at once (image (r&1) was last read for row r-2, before the barrier of row r-1)
                    if (store_row && valid) {
#pragma unroll
                        for (int x = 0; x < TX; ++x) st[x * D + d] = v[x];
                    }
                    if constexpr (FUSE && !(ROLES && DFE_LEAD_FROM_IMAGE)) {   // centre cell and lead cells leave from registers; the minimum comes from the image
                        if (store_row) {
                            int lf = lane;
                            asm volatile("" : "+v"(lf));
                            // (33 x 33: the centre cell, 1-based 545, is lane 32 of chunk 8)
                            fuse_plain_stores<TX>(v, lf, wave, pg_run, fa, nover, CG ? 8 : fa.cmid, CG ? 32 : fa.lmid);
                        }
                    }
                }
            };
            auto do_quarter = [&]() {
                if (has_q) {
                    // the quarter's frame-0 window always comes through LDS + DPP: its column offset is a run-time value and a
                    // second set of 24 scalars next to the main task's 42 does not fit the SGPR file
                    const px_t a = t0[t0r + TQ * (wave - QW0) + l16];
                    const float avq[C][TQ + K - 1] = {};
                    float v[TQ];
                    rowimg_task_row<C, K, TQ, m, false>(row_ptr(lpq, r), a, avq, ringq, v);
                    if constexpr (SWEEP) ring_step(lpq);
                    if (store_row && validq) {
                        const int dqf = 1024 + lane_id_fresh();
#pragma unroll
                        for (int x = 0; x < TQ; ++x) st[(TQ * (wave - QW0) + x) * D + dqf] = v[x];
                    }
                } else if constexpr (K == 7 && DFE_Q_UNDEF) {
                    // the other waves never read their quarter state: "redefine" the two slots the task would have written, so that the join
                    // of the two paths needs no register copies on this side (4 v_mov per row on 12 waves otherwise)
#pragma unroll
                    for (int x = 0; x < TQ; ++x) asm volatile("" : "=v"(ringq[(m + 5) % 6][x]), "=v"(ringq[m][x]));
                }
            };
            auto do_mini = [&]() {
                if (has_m) {
                    // per-lane frame-0 pixels: no broadcast here, every lane has its own column
                    const px_t *lr = row_ptr(lpm, r);
                    if constexpr (SWEEP) ring_step(lpm);
                    const px_t *ar = t0 + t0r + xm;
                    float r0 = 0.f, r2 = 0.f, r4 = 0.f, r5 = 0.f;
                    if constexpr (K == 7) {   // the box-filter state, in flight with the first pixel reads
                        r0 = rm[m * 64], r2 = rm[((m + 2) % 6) * 64], r4 = rm[((m + 4) % 6) * 64], r5 = rm[((m + 5) % 6) * 64];
                    }
                    float e[K];
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        if (j % 4 == 0) __builtin_amdgcn_sched_barrier(0);   // two groups: at most 8 pixel reads in flight (registers)
                        const px_t av = ar[j], bv = lr[j];
                        if constexpr (C == 1) {
                            const float a1[1] = {av};
                            e[j] = sqdiff<1>(a1, bv);
                        } else {
                            const float a3[3] = {av.x, av.y, av.z};
                            e[j] = sqdiff<3>(a3, bv);
                        }
                    }
                    // right-to-left chain = the association of column 0 of a main task (hsum_vh: sa[0]): the tail cells this task
                    // computes for the pixel after the run are bit-identical to what that pixel's own tile holds in its image
                    float h[1], v;
                    h[0] = e[K - 1];
#pragma unroll
                    for (int j = K - 2; j >= 0; --j) h[0] = e[j] + h[0];
                    if constexpr (K == 7) {
                        v = (r0 + r2) + (r4 + h[0]);
                        rm[((m + 5) % 6) * 64] = r5 + h[0];
                        rm[m * 64] = h[0];
                    } else {
                        rm[m * 64] = h[0];
                        v = rm[((m + 1) % K) * 64];
#pragma unroll
                        for (int i = 2; i <= K; ++i) v += (i == K) ? h[0] : rm[((m + i) % K) * 64];
                    }
                    if (store_row) {
                        int lm = lane;
                        asm volatile("" : "+v"(lm));   // (keeps this address and the lane masks out of the scalar file / the spill slots)
                        const bool mt = TOWN && lm >= 32;
                        // (tail cells: only those that complete the run's last line, a0 + RUN + ntl == 0 mod 32)
                        const bool ok = mt ? has_next && lm - 32 < ((-(a0 + RUN)) & 31) : 1088 + (lm >> 3) < D && (!TOWN || lm < 32);
                        if (ok) st[mt ? TX * D + (lm - 32) : (lm & 7) * D + 1088 + (lm >> 3)] = v;
                    }
                }
            };
            // (order of a wave's tasks within a row: the latency-bound extra tasks in front of the main task -- so that they run
            //  while the SIMD's other waves are busy, instead of alone behind them -- measured +-1 %)
            DFE_MARK("main");
            do_main();
            DFE_MARK("quarter");
            do_quarter();
            DFE_MARK("mini");
            do_mini();
            DFE_MARK("barrier");
            if (SWEEP || store_row) {
                // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for the previous row's
                // global stores to be acknowledged before every barrier and serialise stores with compute.
                // One barrier per row is enough with two images: a wave re-deposits into image (r&1) only
                // after the barrier of row r+1, which every wave reaches after its copy-out of row r.
                // (The column sweep needs it in the warm-up rows too: it also frees the tile row the sweep has just left.)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                DFE_MARK("refill");
                if constexpr (SWEEP) {
                    if (wave == LW) {
                        // Stream the rings: the pixels requested one row step ago go into the slots of the rows the sweep
                        // left at r-1 (no wave reads them again; their first use is >= 2 barriers away), then the rows
                        // that will take the slots of row r are requested.  This wave issues no stores at all, so waiting
                        // for its loads never waits for the store stream (loads and stores share vmcnt on gfx9).
                        px_t *t0w = reinterpret_cast<px_t *>(dfe_smem + g_tile0_off);
                        int lw = lane;
                        asm volatile("" : "+v"(lw));
                        const bool t1lane = lw < g_lcols, t0lane = !t1lane && lw - g_lcols <= NE;
                        // (the requests are DFE_REFILL_AHEAD = 2 row steps old when they are waited for: one step -- 1.3 us -- is
                        //  about a memory round trip behind this kernel's own store stream, and the wait is on every wave's path
                        //  to the next barrier)
                        constexpr int RA = DFE_REFILL_AHEAD, hs = m % RA;
                        static_assert(VUnroll<K>::value % RA == 0, "hold slot = row step mod RA must be a compile-time value");
                        if (r >= RA) {
                            px_t px;
                            if constexpr (C == 1) px = hold[hs][0]; else px = make_float4(hold[hs][0], hold[hs][1], hold[hs][2], 0.f);
                            if (t1lane) lds[((r - RA) & (g_lrows - 1)) * g_pitch + lw] = px;
                            if (t0lane) t0w[((r - RA) & (R0 - 1)) * kT0W + lw - g_lcols] = px;
                        }
                        // (running row pointers instead of these multiplies -- one add per row, fewer scalars to keep -- measured 4 %
                        //  slower on the plain build and changed nothing in the spill count)
                        const float *src = t1lane ? I1 + (long long)min(y0 + r + g_lrows, p.H - 1) * p.W + x0 + lw
                                                  : I0 + (long long)min(y0 + oy + r + R0, p.H - 1) * p.W + x0 + ox + (lw - g_lcols);
                        if (t1lane || t0lane) {
#pragma unroll
                            for (int c = 0; c < C; ++c) hold[hs][c] = src[c * HW];
                        }
                    }
                }
                DFE_MARK("record+scan");
                if constexpr (ROLES) {
                    // (row r-1's record is complete: every scan wave has passed this barrier; its buffer is rewritten by the scan of
                    //  row r+1, behind the next barrier)
                    // (written by the first COPY wave: wave LW refills the rings and waits for its loads with vmcnt(0) -- a store of its own
                    //  would put the whole store stream's latency into that wait: 1080p 2.32 against 1.8 ms)
                    if (wave == TX && r - 1 >= K - 1 && y - 1 >= y0n) write_record((r - 1) & 1, rec_prev);
                    if (store_row && wave < TX) scan_row_whole(st, r & 1);
                } else if constexpr (FUSE && !DFE_SCAN_AFTER_COPY) {
                    if (store_row) scan_row(st, pg_run);
                }
                DFE_MARK("copy");
                // Copy-out by the waves WITHOUT an extra task (DFE_CW0.., 10 or 11 of them): a CU's vector-memory path takes 64 B
                // per clock, i.e. ~545 cycles for the 34 848 B of a row, and every wave that stores waits its turn in it.  With all
                // waves copying, the four quarter-task waves -- the critical path of the sweep -- started the next row up to 0.2 us
                // late (per-CU row step 1.48 us against 1.27 us of compute, measured with the memory unsaturated); the copier waves
                // have that much slack before the next barrier.
                // (Fused build, one channel or static tiles: all waves copy -- they are all held by the scan before it, and fewer
                //  copiers then only take longer: 298 against 290 us.  Fused 3-channel sweep: ROLES above -- the copy waves are 8..14;
                //  6 / 5 / 4 of them measured 269 / 277 / 297 against 258 us, scan waves taking one or two pieces each 289 / 294 us,
                //  only the four scan waves without a quarter task taking them 270 us.)
                constexpr int CW0 = ROLES ? TX : FUSE ? 0 : DFE_CW0;          // first copier wave
                constexpr int NCW = (HAS_XW ? LW : NW) - CW0;                // copier waves
                if (store_row && wave >= CW0 && (!HAS_XW || wave != LW)) {
                    const int ov = nover * D;                                // floats of the run that are the neighbour's (shifted last tile)
                    int tj = tid - CW0 * 64;
                    asm volatile("" : "+v"(tj));   // keeps per-lane copy addresses from being hoisted (and spilled)
                    constexpr int STR = NCW * 64;
                    int head, nbody4;
                    if constexpr (F16) {
                        // 8 cells per piece: two b128 reads of the fp32 image -> v_cvt_pkrtz would truncate, so four
                        // v_cvt_f16_f32 pairs (round to nearest even) packed into one dwordx4 store
                        head = (64 - (a0 + ov)) & 63;                        // cells before the first whole line
                        const int nbody8 = ((RUN - ov - head) >> 6) << 3;    // 8-cell pieces in whole 128-B lines
                        nbody4 = nbody8 << 1;                                // (in 4-cell units, for the tail below)
                        const f4_t *sb = reinterpret_cast<const f4_t *>(st + ov + head);
                        const _Float16 *gb = reinterpret_cast<const _Float16 *>(reinterpret_cast<const char *>(out) + G0b) + (ov + head);
                        constexpr int NPC = ((TX * 1096 + 64) / 8 + STR - 1) / STR;
                        const float sc = p.scale;
                        f4_t lo[NPC], hi[NPC];
#pragma unroll
                        for (int i = 0; i < NPC; ++i) {
                            const int j = min(tj + i * STR, nbody8 - 1);
                            lo[i] = sb[2 * j];
                            hi[i] = sb[2 * j + 1];
                        }
#pragma unroll
                        for (int i = 0; i < NPC; ++i)
                            if (tj + i * STR < nbody8) {
                                typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                                union { h2_t h[4]; f4_t v; } u;
                                u.h[0] = h2_t{(_Float16)(lo[i][0] * sc), (_Float16)(lo[i][1] * sc)};
                                u.h[1] = h2_t{(_Float16)(lo[i][2] * sc), (_Float16)(lo[i][3] * sc)};
                                u.h[2] = h2_t{(_Float16)(hi[i][0] * sc), (_Float16)(hi[i][1] * sc)};
                                u.h[3] = h2_t{(_Float16)(hi[i][2] * sc), (_Float16)(hi[i][3] * sc)};
                                asm volatile("global_store_dwordx4 %0, %1, %2" DFE_ST_FLAGS ::"v"((unsigned)(tj + i * STR) * 16u), "v"(u.v), "s"(gb) : "memory");
                            }
                    } else {
                    head = (32 - (a0 + ov)) & 31;                            // floats before the first whole line
                    const int ntl = has_next ? (-(a0 + RUN)) & 31 : 0;       // cells of the next pixel that complete the last line
                    nbody4 = ((RUN - ov - head + ntl) >> 5) << 3;            // float4 pieces in whole 128-B lines
                    const f4_t *sb = reinterpret_cast<const f4_t *>(st + ov + head);
                    const float *gb = reinterpret_cast<const float *>(reinterpret_cast<const char *>(out) + G0b) + (ov + head);
                    // at most 3 (4 with 10 copier waves) pieces per thread: all LDS reads first, then the
                    // stores -- one exposed LDS latency instead of three (every wave of the block is in this phase at once,
                    // nothing else hides it: -6 % on the whole kernel).  Tried on top: the scan's reads issued together with
                    // these (scan arithmetic after the stores: +3 %; before them: no change).
                    constexpr int NPC = ((TX * 1096 + 32) / 4 + STR - 1) / STR;   // pieces per thread (3 with 15 copier waves, 4 with 10)
                    constexpr int GP = NPC <= 4 ? NPC : 6;                        // pieces in flight per thread
                    // A run that is stored whole (every tile but a shifted last one) has at least ((8 * 1089 - 31) >> 5) << 3 = 2168 pieces in whole
                    // lines: the first NFULL pieces of every thread are inside whatever the run's alignment.  Those need no clamp, no compare and
                    // no execution mask, and their addresses are ONE per-lane offset (16 tj) next to per-piece LDS immediates and per-piece scalar
                    // bases (SALU) -- 2 vector instructions per thread and row instead of 5 per piece (round 4: the copy waves' address
                    // arithmetic was 30 of their 55 VALU per row, tools/isa_regions.py).
                    constexpr int NFULL = (DFE_COPY_FAST && CG && !F16) ? ((2168 / STR < NPC - 1) ? 2168 / STR : NPC - 1) : 0;
                    if (NFULL > 0 && nover == 0) {   // block-uniform
                        const unsigned vo = (unsigned)tj * 16u;
                        const char *sbb = reinterpret_cast<const char *>(sb) + vo;
                        const char *gbb = reinterpret_cast<const char *>(gb);
                        f4_t val[NPC];
#pragma unroll
                        for (int i = 0; i < NFULL; ++i) val[i] = *reinterpret_cast<const f4_t *>(sbb + i * STR * 16);
#pragma unroll
                        for (int i = NFULL; i < NPC; ++i) val[i] = sb[min(tj + i * STR, nbody4 - 1)];
#pragma unroll
                        for (int i = 0; i < NFULL; ++i)
                            asm volatile("global_store_dwordx4 %0, %1, %2" DFE_ST_FLAGS ::"v"(vo), "v"(val[i]), "s"(gbb + (size_t)i * STR * 16) : "memory");
#pragma unroll
                        for (int i = NFULL; i < NPC; ++i)
                            if (tj + i * STR < nbody4)
                                asm volatile("global_store_dwordx4 %0, %1, %2" DFE_ST_FLAGS ::"v"((unsigned)(tj + i * STR) * 16u), "v"(val[i]), "s"(gb) : "memory");
                    } else
#pragma unroll
                    for (int g0 = 0; g0 < NPC; g0 += GP) {
                        f4_t val[GP];
#pragma unroll
                        for (int i = 0; i < GP; ++i)
                            if (g0 + i < NPC) val[i] = sb[min(tj + (g0 + i) * STR, nbody4 - 1)];
#pragma unroll
                        for (int i = 0; i < GP; ++i)
                            if (g0 + i < NPC)
                                if (tj + (g0 + i) * STR < nbody4)
                                    asm volatile("global_store_dwordx4 %0, %1, %2" DFE_ST_FLAGS ::"v"((unsigned)(tj + (g0 + i) * STR) * 16u), "v"(val[i]), "s"(gb) : "memory");
                    }
                    }
                    {     // the run's two partial lines: head by wave 5, tail by wave 6
                        const int tail0 = ov + head + (nbody4 << 2), ntail = RUN - tail0;   // (tail line owned: ntail <= 0)
                        if constexpr (F16) {
                            _Float16 *oh = reinterpret_cast<_Float16 *>(reinterpret_cast<char *>(out) + G0b);
                            if (wave == CW0 && lane < head) oh[ov + lane] = (_Float16)(st[ov + lane] * p.scale);
                            if (wave == CW0 + 1 && lane < ntail) oh[tail0 + lane] = (_Float16)(st[tail0 + lane] * p.scale);
                        } else {
                            float *og = reinterpret_cast<float *>(reinterpret_cast<char *>(out) + G0b);
                            if (wave == CW0 && lane < head && !skip_head) og[ov + lane] = st[ov + lane];
                            if (wave == CW0 + 1 && lane < ntail) og[tail0 + lane] = st[tail0 + lane];
                        }
                    }
                }
                if constexpr (FUSE && DFE_SCAN_AFTER_COPY) {
                    if (store_row) scan_row(st, pg_run);
                }
            }
            DFE_MARK("rowend");
            if constexpr (ROLES) rec_prev += DFE_REC;
        });
    }
    if constexpr (ROLES) {   // the piece's last row: its scan has no next barrier to be written behind
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int rl = nsweep - 1, yl = y0 + rl - (K - 1);
        if (wave == TX && rl >= K - 1 && yl >= y0n) write_record(rl & 1, rec_prev);   // (rec_prev has been stepped past the last row: its record)
    }
    if constexpr (!SWEEP) break;
    pos += prows;
    if (pos >= pend) break;
    // next piece: every wave is past its last reads of the rings and the images (LDS-only barrier, the stores drain on)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
    // (Round 4 tried the finalize INSIDE this kernel: at the end of the launch each block finished the pixels of its own tile rows --
    //  centre override, decode, extractOutput, depth, its share of the frame border -- with the code flow_finalize_kernel runs
    //  (dfe_finalize_rec_pixel), bit-identical, no second launch.  Same-call A/B, profiles/r04_ac_ab_finalize_in_sweep.txt: step
    //  0.2486 / 0.2508 / 0.2504 ms against 0.2496 / 0.2483 / 0.2509 ms with the launch -- the kernel grew by the 5.5 us that the phase takes
    //  behind the drain of the block's stores (two dependent passes over ~1100 pixels a block), which is what the launch and its boundary
    //  cost: a step is kernel + ~6 us of launch boundary whether that holds one kernel or two.  Taken out again.)
}

// LDS bytes of a static-tile block of `ty` output rows (0 = does not apply) and the kernel arguments that go with it
template <int C, int K, int TX>
static size_t rowimg_plan(int ty, int H, int W, long long plane, int hWin, int wWin, CvTiledArgs *out_args, bool f16 = false) {
    using px_t = typename Px<C>::type;
    constexpr int U = VUnroll<K>::value;
    const int Ho = H - K + 1 - hWin + 1, Wo = W - K + 1 - wWin + 1;
    const int D = hWin * wWin;
    // 16 chunk waves + the quarter/mini tasks cover up to 1096 cells; below 13 chunks too many waves would idle
    const int rows = ty + K - 1;
    if (D <= 768 || D > 1096 || ty < 1 || Ho < ty || Wo < TX || rows % U != 0) return 0;   // (the kernel sweeps whole groups of U rows)
    CvTiledArgs a;
    a.plane = plane;
    a.H = H; a.W = W; a.hWin = hWin; a.wWin = wWin; a.Ho = Ho; a.Wo = Wo;
    a.lrows = rows + hWin - 1;
    a.lcols = TX + K - 1 + wWin - 1;
    const int M = Px<C>::bank_mod;
    a.pitch = a.lcols;
    while ((a.pitch - wWin) % M != 0) ++a.pitch;
    a.seg_rows = ty;
    size_t tile_bytes = (size_t)a.lrows * a.pitch * sizeof(px_t);
    a.tile0_off = (int)((tile_bytes + 127) / 128 * 128);
    a.stage_off = a.tile0_off + (int)(((size_t)rows * kT0W * sizeof(px_t) + (size_t)U * 64 * sizeof(float) + 127) / 128 * 128);
    a.stage_len = (TX * D + (f16 ? 64 : 32) + 31) / 32 * 32;
    a.chunk0 = 0;
    a.sw_ovh = a.sw_min = 0;
    a.scale = 1.f;
    size_t lds_bytes = a.stage_off + (size_t)2 * a.stage_len * sizeof(float) + 2 * DFE_REC * sizeof(float);   // images + the fused tile-row records
    if (lds_bytes > 160 * 1024) return 0;
    if (out_args) *out_args = a;
    return lds_bytes;
}

// static tiles of `ty` output rows
template <int C, int K, int TX, bool FUSE, bool F16 = false>
static int launch_cv_rowimg_one(dfe_ctx *ctx, const float *I0, const float *I1, int H, int W, long long plane, int hWin, int wWin,
                                int ty, float *out, const CvFuseArgs *fa, bool *handled, float scale = 1.f) {
    *handled = false;
    CvTiledArgs a;
    const size_t lds_bytes = rowimg_plan<C, K, TX>(ty, H, W, plane, hWin, wWin, &a, F16);
    if (!lds_bytes) return DFE_OK;
    a.scale = scale;
    // the +-16 search (33 x 33 = 1089 cells) gets the instantiation with D as a constant
    if (hWin == 33 && wWin == 33 && (a.pitch != RowimgGeom<C, K, TX>::pitch33 ||
                                     a.stage_len != (F16 ? RowimgGeom<C, K, TX>::stage_len33h : RowimgGeom<C, K, TX>::stage_len33)))
        return dfe_fail(ctx, DFE_E_UNSUPPORTED, "row-image kernel: LDS geometry differs from the kernel's constants");
    auto kern = (hWin == 33 && wWin == 33) ? ssd_cv_rowimg_kernel<C, K, TX, DFE_RI_SMEM, FUSE, false, 1089, F16>
                                      : ssd_cv_rowimg_kernel<C, K, TX, DFE_RI_SMEM, FUSE, false, 0, F16>;
    DFE_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    dim3 grid(dfe_cdiv(a.Wo, TX), dfe_cdiv(a.Ho, ty));
    {
        DfeProfScope prof(ctx, true);
        hipExtLaunchKernelGGL(kern, grid, dim3(1024), lds_bytes, ctx->stream, prof.a, prof.b, 0, I0, I1, out, a, fa ? *fa : CvFuseArgs{});
    }
    DFE_LAUNCH_CHECK(ctx);
    ctx->last_kernel = F16 ? (FUSE ? "ssd_cv_rowimg_kernel_f16+fused_tail" : "ssd_cv_rowimg_kernel_f16")
                           : (FUSE ? "ssd_cv_rowimg_kernel+fused_tail" : "ssd_cv_rowimg_kernel");
    *handled = true;
    return DFE_OK;
}

// Column sweep, persistent: one block per CU, each walks its share of the column-major (8-pixel column, row) sequence
// (sweep_cut).  Row steps per block ~ (ncols*Ho + ovh*(ncols + B)) / B -- VGA: 141 instead of the 3 x 51 of a 76 x 10 grid
// of one-segment blocks; every CU ends within one row step of the others.
constexpr int kSweepOvh = 9;    // row steps a piece costs before its first stored row: K-1 = 6 warm-up rows + ~3 for staging the rings
constexpr int kSweepMin = 8;    // no piece shorter than this
// fused build: cost of a swept row of the sweep relative to static tiles.  One channel (all waves scan and copy): 2.05 against
// 1.88 us at VGA.  Three channels (scan / copy roles in both forms): 1.62 against 1.66 us.
constexpr double kSweepFusedPenalty1 = 1.08, kSweepFusedPenalty3 = 1.0;
static double sweep_cost(int ncols, int Ho, int B) { return ((double)ncols * Ho + (double)kSweepOvh * (ncols + B)) / B; }
static int sweep_blocks(const dfe_ctx *ctx, int ncols, int Ho) {
    long long b = (long long)ncols * Ho / 24;          // at least ~24 rows of work per block
    if (b > ctx->ncu) b = ctx->ncu;
    return b < 1 ? 1 : (int)b;
}
// Aligned fronts: with k = CUs / columns >= 2 whole segments per column, k * ncols blocks cut every column into k equal
// segments (ovh = 0 in sweep_cut), so the blocks of one segment index walk down their columns in lockstep and write a
// CONTIGUOUS image row (ncols x 34 KB) at a time -- k moving fronts instead of one per CU.  The memory system prefers that
// by far (tools/ubench/stores11.hip, whole-line nt stores paced like the kernel: 3 fronts of 76 runs 6.1 TB/s, 256
// scattered 34-KB fronts 4.7-5.3 TB/s, one contiguous 8.9-MB front 6.6 TB/s), and the plain build is bound by its store
// stream: VGA, 228 aligned blocks x 154 row steps measures the same as or better than 256 balanced blocks x 141 (222 / 221 us,
// 228 / 238 us on two boxes).  The fused build is bound by its arithmetic and keeps the balanced cut.
static int sweep_aligned_k(const dfe_ctx *ctx, int ncols, int Ho) {
    const int k = ctx->ncu / ncols;
    return (k >= 2 && Ho / k >= 24) ? k : 0;
}
template <int C, int K, int TX, bool FUSE>
static int launch_cv_rowimg_sweep(dfe_ctx *ctx, const float *I0, const float *I1, int H, int W, long long plane, int hWin, int wWin,
                                  float *out, const CvFuseArgs *fa, bool *handled) {
    using px_t = typename Px<C>::type;
    constexpr int U = VUnroll<K>::value;
    constexpr int R = 64, R0 = 8;                 // ring rows (frame-1: a power of two >= hWin + 2; frame-0)
    const int Ho = H - K + 1 - hWin + 1, Wo = W - K + 1 - wWin + 1;
    const int D = hWin * wWin;
    *handled = false;
    if (D <= 768 || D > 1096 || Ho < 1 || Wo < TX || hWin + 2 > R || (hWin - 1) / 2 + R0 > hWin - 1) return DFE_OK;
    CvTiledArgs a;
    a.plane = plane;
    a.H = H; a.W = W; a.hWin = hWin; a.wWin = wWin; a.Ho = Ho; a.Wo = Wo;
    a.lrows = R;
    a.lcols = TX + K - 1 + wWin - 1;
    if (a.lcols + TX + K > 64) return DFE_OK;   // one wave streams a tile row and a frame-0 row (NE + 1 pixels) per step
    const int M = Px<C>::bank_mod;
    a.pitch = a.lcols;
    while ((a.pitch - wWin) % M != 0) ++a.pitch;
    const int ncols = dfe_cdiv(Wo, TX);
    a.seg_rows = 0;
    a.scale = 1.f;
    a.sw_ovh = kSweepOvh; a.sw_min = kSweepMin;
    int nblk = sweep_blocks(ctx, ncols, Ho);
    if (const int k = FUSE ? 0 : sweep_aligned_k(ctx, ncols, Ho)) { nblk = k * ncols; a.sw_ovh = 0; }
    if (ctx->opt[DFE_OPT_SWEEP_OVH] >= 0) a.sw_ovh = ctx->opt[DFE_OPT_SWEEP_OVH];      // tuning
    if (ctx->opt[DFE_OPT_SWEEP_BLOCKS] >= 0) nblk = ctx->opt[DFE_OPT_SWEEP_BLOCKS];    // tuning
    if (nblk < 1 || (long long)(Ho + a.sw_ovh) * ncols * nblk >= (1ll << 31)) return DFE_OK;   // sweep_cut works in 32 bits
    size_t tile_bytes = (size_t)a.lrows * a.pitch * sizeof(px_t);
    a.tile0_off = (int)((tile_bytes + 127) / 128 * 128);
    a.stage_off = a.tile0_off + (int)(((size_t)R0 * kT0W * sizeof(px_t) + (size_t)U * 64 * sizeof(float) + 127) / 128 * 128);
    a.stage_len = (TX * D + 32 + 31) / 32 * 32;
    a.chunk0 = 0;
    size_t lds_bytes = a.stage_off + (size_t)2 * a.stage_len * sizeof(float) + 2 * DFE_REC * sizeof(float);   // images + the fused tile-row records
    if (lds_bytes > 160 * 1024) return DFE_OK;
    const bool sq33 = hWin == 33 && wWin == 33;
    if (sq33 && (a.pitch != RowimgGeom<C, K, TX>::pitch33 || a.tile0_off != RowimgGeom<C, K, TX>::sweep_tile0_off ||
                 a.stage_off != RowimgGeom<C, K, TX>::sweep_stage_off || a.stage_len != RowimgGeom<C, K, TX>::stage_len33))
        return dfe_fail(ctx, DFE_E_UNSUPPORTED, "row-image sweep: LDS geometry differs from the kernel's constants");
    // the fused sweep fits the register file only with D as a constant (33 x 33); other windows: static tiles
    if (FUSE && !sq33) return DFE_OK;
    auto kern = sq33 ? ssd_cv_rowimg_kernel<C, K, TX, DFE_RI_SMEM, FUSE, true, 1089>
                            : ssd_cv_rowimg_kernel<C, K, TX, DFE_RI_SMEM, false, true>;
    DFE_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    dim3 grid(nblk, 1);
    {
        DfeProfScope prof(ctx, true);
        hipExtLaunchKernelGGL(kern, grid, dim3(1024), lds_bytes, ctx->stream, prof.a, prof.b, 0, I0, I1, out, a, fa ? *fa : CvFuseArgs{});
    }
    DFE_LAUNCH_CHECK(ctx);
    ctx->last_kernel = FUSE ? "ssd_cv_rowimg_kernel+fused_tail" : "ssd_cv_rowimg_kernel";
    *handled = true;
    return DFE_OK;
}

// Static tile height for a frame: the kernel's time follows rounds x rows swept -- blocks / CUs, rounded up, times the
// ty + K-1 rows a block sweeps (the K-1 warm-up rows are the overhead of short tiles, the under-full last round that of
// a block count that does not fit the CU count).  VGA, 33 x 33: ty = 48 -> 10 x 76 = 760 blocks = 3 rounds x 54 rows,
// ty = 36 -> 988 blocks = 4 x 42, ty = 24 -> 1444 blocks = 6 x 30; measured (fused build) 304 / 309 / 324 us.
// (With ordinary instead of non-temporal copy-out stores the height made no difference at all: the store stream was
// the co-bottleneck, and a short tile's warm-up rows overlapped the drain of the previous tile's stores.)
// ty + K-1 is a multiple of the row unroll U; LDS (the frame-1 tile grows with ty) allows up to 48 rows at 33 x 33.
template <int C, int K, int TX>
static int rowimg_pick_ty(const dfe_ctx *ctx, int H, int W, long long plane, int hWin, int wWin, double *cost_out, bool f16 = false) {
    constexpr int U = VUnroll<K>::value;
    const int Ho = H - K + 1 - hWin + 1, Wo = W - K + 1 - wWin + 1;
    const int ncols = dfe_cdiv(Wo, TX);
    int best = 0;
    double best_cost = 1e30;
    for (int ty = U - (K - 1) % U; ty <= Ho && ty <= 20 * U; ty += U) {
        if (ty < 1 || !rowimg_plan<C, K, TX>(ty, H, W, plane, hWin, wWin, nullptr, f16)) continue;
        const double cost = (double)dfe_cdiv((long long)ncols * dfe_cdiv(Ho, ty), ctx->ncu) * (ty + K - 1);
        if (cost < best_cost) { best_cost = cost; best = ty; }   // ties: the shorter tile (less LDS, finer last round)
    }
    if (cost_out) *cost_out = best_cost;
    return best;
}

// row-image kernel.  dfe_set_cost_volume_tile: 0 = auto (static tiles of the height rowimg_pick_ty chooses, or the column
// sweep, see below), 1 = force the column sweep, 3..7 = static tiles of 6n - 6 rows, 100 + ty = static tiles of ty rows
// (ty + K-1 a multiple of the row unroll)
template <int C, int K, int TX, bool FUSE>
static int launch_cv_rowimg(dfe_ctx *ctx, const float *I0, const float *I1, int H, int W, long long plane, int hWin, int wWin,
                            float *out, const CvFuseArgs *fa, bool *handled) {
    constexpr int U = VUnroll<K>::value;
    const int code = ctx->cv_tyq;
    *handled = false;
    if (code == 1)     // forced column sweep (fused: 33 x 33 windows only)
        return launch_cv_rowimg_sweep<C, K, TX, FUSE>(ctx, I0, I1, H, W, plane, hWin, wWin, out, fa, handled);
    int ty;
    if (code >= 100) ty = code - 100;
    else if (code >= 2) ty = U * code - (K - 1);
    else {
        // Static tiles or the persistent column sweep?  Both cost about the same per swept row in the plain build; the sweep pays
        // the K-1 warm-up rows once per piece and ends all CUs together, so the smaller count of row steps per CU wins.  In the
        // fused one-channel build a swept row of the sweep costs more (kSweepFusedPenalty1).
        const int Ho = H - K + 1 - hWin + 1, Wo = W - K + 1 - wWin + 1;
        double cost_static = 1e30;
        ty = rowimg_pick_ty<C, K, TX>(ctx, H, W, plane, hWin, wWin, &cost_static);
        if (ty && Wo >= TX) {
            const int ncols = dfe_cdiv(Wo, TX);
            const int ka = FUSE ? 0 : sweep_aligned_k(ctx, ncols, Ho);
            const double cost_sweep = ka ? dfe_cdiv(Ho, ka) + K - 1 : sweep_cost(ncols, Ho, sweep_blocks(ctx, ncols, Ho));
            if (cost_sweep * (FUSE ? (C == 3 ? kSweepFusedPenalty3 : kSweepFusedPenalty1) : 1.0) < cost_static) {
                int rc = launch_cv_rowimg_sweep<C, K, TX, FUSE>(ctx, I0, I1, H, W, plane, hWin, wWin, out, fa, handled);
                if (rc != DFE_OK || *handled) return rc;
            }
        }
    }
    return launch_cv_rowimg_one<C, K, TX, FUSE>(ctx, I0, I1, H, W, plane, hWin, wWin, ty, out, fa, handled);
}

// fp16 volume (cost * scale): static tiles of the row-image kernel, the height rowimg_pick_ty chooses for the (slightly larger) image
template <int C, int K, int TX, bool FUSE>
static int launch_cv_rowimg_f16(dfe_ctx *ctx, const float *I0, const float *I1, int H, int W, long long plane, int hWin, int wWin, float scale,
                                void *out, const CvFuseArgs *fa, bool *handled) {
    *handled = false;
    int ty;
    const int code = ctx->cv_tyq;
    if (code >= 100) ty = code - 100;
    else ty = rowimg_pick_ty<C, K, TX>(ctx, H, W, plane, hWin, wWin, nullptr, true);
    if (ty < 1) return DFE_OK;
    return launch_cv_rowimg_one<C, K, TX, FUSE, true>(ctx, I0, I1, H, W, plane, hWin, wWin, ty, (float *)out, fa, handled, scale);
}

// the row-image kernel's role-split instantiations (3 channels, 33 x 33, fused) leave per-pixel RECORDS (fa.rec) instead of the planes
static bool rowimg_writes_records(int C, int hWin, int wWin) { return DFE_ROLES && DFE_ROLES_STATIC && DFE_LEAD_FROM_IMAGE && C == 3 && hWin == 33 && wWin == 33; }

// fp16 build of raw frames (C in {1, 3}, k = 7, 769..1096 window cells); *handled = false: no fast kernel for the shape
int cv_frames_dispatch_f16(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, long long plane, int k, int hWin, int wWin,
                           float scale, void *out, const CvFuseArgs *fa, bool *handled, bool *recs = nullptr) {
    *handled = false;
    if (recs) *recs = fa && rowimg_writes_records(C, hWin, wWin);
    if (ctx->cv_mode == 1 || ctx->cv_mode == 2 || k != 7 || (C != 3 && C != 1)) return DFE_OK;
    if (fa) return C == 3 ? launch_cv_rowimg_f16<3, 7, 8, true>(ctx, I0, I1, H, W, plane, hWin, wWin, scale, out, fa, handled)
                          : launch_cv_rowimg_f16<1, 7, 8, true>(ctx, I0, I1, H, W, plane, hWin, wWin, scale, out, fa, handled);
    return C == 3 ? launch_cv_rowimg_f16<3, 7, 8, false>(ctx, I0, I1, H, W, plane, hWin, wWin, scale, out, nullptr, handled)
                  : launch_cv_rowimg_f16<1, 7, 8, false>(ctx, I0, I1, H, W, plane, hWin, wWin, scale, out, nullptr, handled);
}

// fp32 -> fp16 with the volume's scale (shapes without a fast fp16 kernel: built in fp32 bands, converted)
__global__ __launch_bounds__(256) void cv_to_half_kernel(const float *__restrict__ in, long long n, float scale, _Float16 *__restrict__ out) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) out[e] = (_Float16)(in[e] * scale);
}

// volumes of one call above this size leave the tiled multi kernel with the non-temporal hint (they will not be found in the
// 256-MB memory-side cache by the cascade kernels anyway; below it the cascade's reads hit what plain stores left there)
constexpr size_t kXposeNtBytes = (size_t)160 << 20;
// the volumes of n independent frame pairs with windows of at most one chunk (C = 3, k = 7) in one launch; *handled = false
// when some pair has no plan with the common block shape (the caller then launches them one by one)
template <int NQ>
static int launch_cv_tiled_multi_one(dfe_ctx *ctx, int n, const float *const *I0, const float *const *I1, const int *H, const int *W,
                                     int hWin, int wWin, float *const *out, float *const *prob, bool *handled, bool *prob_used, float f16_scale) {
    constexpr int C = 3, K = 7, TX = 8, NT = 4, NW = 4;
    CvTiledMulti m;
    size_t lds = 0;
    int gxm = 0, gym = 0;
    for (int i = 0; i < n; ++i) {
        const int Ho = H[i] - K + 1 - hWin + 1, Wo = W[i] - K + 1 - wWin + 1;
        const CvTilePlan pl = plan_cv_tiled<C, K, TX, NT, NW>(NQ, Ho, Wo, hWin, wWin, ctx->ncu);
        if (pl.score <= 0) return DFE_OK;
        CvTiledArgs a;
        a.plane = (long long)H[i] * W[i];
        a.H = H[i]; a.W = W[i]; a.hWin = hWin; a.wWin = wWin; a.Ho = Ho; a.Wo = Wo;
        a.lrows = pl.lrows; a.lcols = pl.lcols; a.pitch = pl.pitch; a.seg_rows = 0; a.tile0_off = 0; a.stage_off = 0; a.stage_len = 0;
        a.chunk0 = 0;
        m.p[i] = a; m.I0[i] = I0[i]; m.I1[i] = I1[i]; m.out[i] = out[i]; m.prob[i] = prob ? prob[i] : nullptr;
        m.gx[i] = dfe_cdiv(Wo, pl.GX); m.gy[i] = dfe_cdiv(Ho, pl.TY);
        if (m.gx[i] > gxm) gxm = m.gx[i];
        if (m.gy[i] > gym) gym = m.gy[i];
        if (pl.lds_bytes > lds) lds = pl.lds_bytes;
    }
    const bool xpose = hWin * wWin == 64 && ctx->opt[DFE_OPT_XPOSE] != 0;
    if (f16_scale != 0.f && !xpose) return DFE_OK;   // (the fp16 store path is the transposed one; the caller converts otherwise)
    if (xpose) {   // a 2-KB transpose scratch per wave behind the largest tile (see the kernel's store path)
        const size_t xoff = (lds + 255) / 256 * 256;
        size_t vol_bytes = 0;
        for (int i = 0; i < n; ++i) {
            vol_bytes += (size_t)m.p[i].Ho * m.p[i].Wo * 64 * (f16_scale != 0.f ? 2 : sizeof(float));
            if (f16_scale != 0.f) { m.p[i].f16 = 1; m.p[i].scale = f16_scale; }
        }
        int nt = vol_bytes > kXposeNtBytes;
        if (ctx->opt[DFE_OPT_XPOSE_NT] >= 0) nt = ctx->opt[DFE_OPT_XPOSE_NT] != 0;   // tuning
        for (int i = 0; i < n; ++i) { m.p[i].stage_off = (int)xoff; m.p[i].stage_len = nt; }
        lds = xoff + (size_t)NW * TX * 64 * sizeof(float);
    }
    // The soft-min epilogue makes a block ~3x longer.  It pays when the first (largest) pair has enough blocks to hide the
    // other pairs' few long ones behind (1080p: 3600 blocks, -5.5 % on the step; 720p: 1600 blocks, -5.6 %); at VGA (540
    // blocks) those long blocks set the launch time and the separate soft-min launch is faster (0.119 against 0.134 ms).
    bool use_prob = prob && m.gx[0] * m.gy[0] >= 1000 && f16_scale == 0.f;
    if (ctx->opt[DFE_OPT_SOFT_EPILOGUE] >= 0) use_prob = prob && ctx->opt[DFE_OPT_SOFT_EPILOGUE] != 0;   // tuning / tests: force on (1) or off (0)
    if (!use_prob)
        for (int i = 0; i < n; ++i) m.prob[i] = nullptr;
    if (prob_used) *prob_used = use_prob;
    auto kern = ssd_cv_tiled_multi_kernel<C, K, TX, NT, NW, NQ>;
    DFE_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        DfeProfScope prof(ctx);
        hipLaunchKernelGGL(kern, dim3(gxm, gym, n), dim3(NW * 64), lds, ctx->stream, m);
    }
    DFE_LAUNCH_CHECK(ctx);
    ctx->last_kernel = f16_scale != 0.f ? "ssd_cv_tiled_multi_kernel_f16" : "ssd_cv_tiled_kernel";
    *handled = true;
    return DFE_OK;
}

// f16_scale != 0: out[i] are HALF volumes, out[i][..] = half(cost * f16_scale) (8 x 8 windows only: *handled = false otherwise)
// the finest pyramid scale through the tiled kernel's fused epilogue (C = 3, k = 7, 8 x 8 windows); I0p / I1p: the padded scale-1 frames
template <int NQ>
static int launch_cv_fine(dfe_ctx *ctx, const float *I0p, const float *I1p, int Hp, int Wp, int maxh, int maxw, const CvFineArgs &fine, bool *handled) {
    constexpr int K = 7, TX = 8, NT = 4, NW = 4;
    const int Ho = Hp - K + 1 - maxh + 1, Wo = Wp - K + 1 - maxw + 1;
    const CvTilePlan pl = plan_cv_tiled<3, K, TX, NT, NW>(NQ, Ho, Wo, maxh, maxw, ctx->ncu);
    if (pl.score <= 0) return DFE_OK;
    CvTiledArgs a;
    a.plane = (long long)Hp * Wp;
    a.H = Hp; a.W = Wp; a.hWin = maxh; a.wWin = maxw; a.Ho = Ho; a.Wo = Wo;
    a.lrows = pl.lrows; a.lcols = pl.lcols; a.pitch = pl.pitch; a.seg_rows = 0; a.tile0_off = 0; a.stage_off = 0; a.stage_len = 0;
    a.chunk0 = 0;
    const bool h16 = fine.f16_scale != 0.f;
    auto kern = fine.casc ? (h16 ? ssd_cv_tiled_fine_kernel<3, K, TX, NT, NW, NQ, true, true> : ssd_cv_tiled_fine_kernel<3, K, TX, NT, NW, NQ, false, true>)
                          : (h16 ? ssd_cv_tiled_fine_kernel<3, K, TX, NT, NW, NQ, true, false> : ssd_cv_tiled_fine_kernel<3, K, TX, NT, NW, NQ, false, false>);
    DFE_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes));
    {
        DfeProfScope prof(ctx);
        hipLaunchKernelGGL(kern, dim3(dfe_cdiv(Wo, pl.GX), dfe_cdiv(Ho, pl.TY)), dim3(NW * 64), pl.lds_bytes, ctx->stream, I0p, I1p, a, fine);
    }
    DFE_LAUNCH_CHECK(ctx);
    ctx->last_kernel = fine.casc ? (h16 ? "ssd_cv_tiled_mid_kernel_f16" : "ssd_cv_tiled_mid_kernel") : h16 ? "ssd_cv_tiled_fine_kernel_f16" : "ssd_cv_tiled_fine_kernel";
    *handled = true;
    return DFE_OK;
}
bool cv_finest_plan_ok(dfe_ctx *ctx, int Hp, int Wp, int maxh, int maxw) {
    if (maxh != 8 || maxw != 8 || ctx->cv_mode == 1 || ctx->cv_mode == 3) return false;
    const int Ho = Hp - 7 + 1 - maxh + 1, Wo = Wp - 7 + 1 - maxw + 1;
    if ((Wo | Ho) & 1) return false;
    return plan_cv_tiled<3, 7, 8, 4, 4>(4, Ho, Wo, maxh, maxw, ctx->ncu).score > 0;
}
int cv_frames_finest_fused(dfe_ctx *ctx, const float *I0p, const float *I1p, int C, int Hp, int Wp, int k, int maxh, int maxw, const CvFineArgs &fine,
                           bool *handled) {
    *handled = false;
    if (C != 3 || k != 7 || maxh != 8 || maxw != 8 || ctx->cv_mode == 1 || ctx->cv_mode == 3) return DFE_OK;
    const int Ho = Hp - 7 + 1 - maxh + 1, Wo = Wp - 7 + 1 - maxw + 1;
    if (fine.pcasc && ((Wo | Ho) & 1)) return DFE_OK;
    if (ctx->opt[fine.casc ? DFE_OPT_MID_NQ : DFE_OPT_FINE_NQ] == 5) return launch_cv_fine<5>(ctx, I0p, I1p, Hp, Wp, maxh, maxw, fine, handled);
    return launch_cv_fine<4>(ctx, I0p, I1p, Hp, Wp, maxh, maxw, fine, handled);
}

int cv_frames_dispatch_multi(dfe_ctx *ctx, int n, const float *const *I0, const float *const *I1, int C, const int *H, const int *W, int k,
                             int hWin, int wWin, float *const *out, float *const *prob, bool *handled, bool *prob_used, float f16_scale, int nq_hint) {
    *handled = false;
    if (prob_used) *prob_used = false;
    if (ctx->cv_mode == 1 || ctx->cv_mode == 3 || C != 3 || k != 7 || hWin * wWin > 64 || n < 2 || n > DFE_MAX_RATIOS) return DFE_OK;
    // 4 row groups (18-row tiles) unless forced: measured at VGA, 3 scales, 2 / 3 / 4 / 5 groups -> 0.161 / 0.159 / 0.151 /
    // 0.155 ms per pair (short tiles pay the K-1 warm-up rows too often, tall ones leave the coarse scales too few blocks)
    // nq_hint: the caller's choice for a launch without the finest scale (the fused pyramid: 3 groups, measured 720p 0.184 -> 0.179,
    // 1080p 0.369 -> 0.363 ms; 4K prefers 4)
    const int nq = (ctx->cv_tyq >= 2 && ctx->cv_tyq <= 5) ? ctx->cv_tyq : (nq_hint >= 2 && nq_hint <= 5) ? nq_hint : 4;
    switch (nq) {
        case 2: return launch_cv_tiled_multi_one<2>(ctx, n, I0, I1, H, W, hWin, wWin, out, prob, handled, prob_used, f16_scale);
        case 5: return launch_cv_tiled_multi_one<5>(ctx, n, I0, I1, H, W, hWin, wWin, out, prob, handled, prob_used, f16_scale);
        case 3: return launch_cv_tiled_multi_one<3>(ctx, n, I0, I1, H, W, hWin, wWin, out, prob, handled, prob_used, f16_scale);
        default: return launch_cv_tiled_multi_one<4>(ctx, n, I0, I1, H, W, hWin, wWin, out, prob, handled, prob_used, f16_scale);
    }
}

// *nparts: planes of fa.part the launched kernel filled -- 2 (the row-image kernel scans a pixel's run in two halves)
// or ceil(D/64) (the tiled kernel leaves one entry per chunk)
int cv_frames_dispatch_fused(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, long long plane, int k, int hWin,
                             int wWin, float *out, const CvFuseArgs &fa, bool *handled, int *nparts, bool *recs) {
    *handled = false;
    if (recs) *recs = false;
    *nparts = (hWin * wWin + 63) / 64;
    if (ctx->cv_mode == 1) return DFE_OK;
    if (hWin * wWin < 64) return DFE_OK;   // less than one full chunk: not worth a fused instantiation
    if ((ctx->cv_mode == 0 || ctx->cv_mode == 3) && (C == 3 || C == 1) && k == 7) {
        int rc = C == 3 ? launch_cv_rowimg<3, 7, 8, true>(ctx, I0, I1, H, W, plane, hWin, wWin, out, &fa, handled)
                        : launch_cv_rowimg<1, 7, 8, true>(ctx, I0, I1, H, W, plane, hWin, wWin, out, &fa, handled);
        if (*handled) *nparts = 2;
        if (*handled && recs) *recs = rowimg_writes_records(C, hWin, wWin);
        if (rc != DFE_OK || *handled) return rc;
    }
    if (ctx->cv_mode == 3) return DFE_OK;   // forced row-image kernel that does not apply: unfused path reports it
    if (C == 3 && k == 7) return launch_cv_tiled_fused<3, 7, 8>(ctx, I0, I1, H, W, plane, hWin, wWin, out, fa, handled);
    if (C == 1 && k == 7) return launch_cv_tiled_fused<1, 7, 8>(ctx, I0, I1, H, W, plane, hWin, wWin, out, fa, handled);
    return DFE_OK;
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
// H is the number of frame rows visible to this call (a row band of a taller frame when plane > H*W)
int cv_frames_dispatch(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, long long plane, int kh,
                              int kw, int hWin, int wWin, float *out) {
    const int Ho = H - kh + 1 - hWin + 1, Wo = W - kw + 1 - wWin + 1;
    if ((ctx->cv_mode == 3 || ctx->cv_mode == 0) && kh == kw && (C == 3 || C == 1) && kh == 7) {
        bool handled = false;
        int rc = C == 3 ? launch_cv_rowimg<3, 7, 8, false>(ctx, I0, I1, H, W, plane, hWin, wWin, out, nullptr, &handled)
                        : launch_cv_rowimg<1, 7, 8, false>(ctx, I0, I1, H, W, plane, hWin, wWin, out, nullptr, &handled);
        if (rc != DFE_OK || handled) return rc;
        if (ctx->cv_mode == 3)
            return dfe_fail(ctx, DFE_E_UNSUPPORTED, "no row-image cost-volume kernel for C=%d k=%d win=%dx%d out=%dx%d", C, kh, hWin, wWin, Ho, Wo);
    }
    if (ctx->cv_mode != 1 && ctx->cv_mode != 3 && kh == kw) {
        bool handled = false;
        int rc = DFE_OK;
        if (C == 3 && kh == 7) rc = launch_cv_tiled<3, 7, 8>(ctx, I0, I1, H, W, plane, hWin, wWin, out, &handled);
        else if (C == 1 && kh == 7) rc = launch_cv_tiled<1, 7, 8>(ctx, I0, I1, H, W, plane, hWin, wWin, out, &handled);
        else if (C == 3 && kh == 5) rc = launch_cv_tiled<3, 5, 8>(ctx, I0, I1, H, W, plane, hWin, wWin, out, &handled);
        else if (C == 3 && kh == 3) rc = launch_cv_tiled<3, 3, 8>(ctx, I0, I1, H, W, plane, hWin, wWin, out, &handled);
        if (rc != DFE_OK || handled) return rc;
    }
    if (ctx->cv_mode == 2)
        return dfe_fail(ctx, DFE_E_UNSUPPORTED, "no tiled cost-volume kernel for C=%d k=%dx%d win=%dx%d out=%dx%d", C, kh,
                        kw, hWin, wWin, Ho, Wo);
    CvRefArgs a;
    a.a = I0; a.a_plane = plane; a.a_pitch = W; a.a_oy = (hWin - 1) / 2; a.a_ox = (wWin - 1) / 2;
    a.b = I1; a.b_plane = plane; a.b_pitch = W;
    a.C = C; a.kh = kh; a.kw = kw; a.Wo = Wo; a.hWin = hWin; a.wWin = wWin;
    a.total = (long long)Ho * Wo * hWin * wWin;
    a.out = out;
    return launch_cv_ref(ctx, a);
}

extern "C" {
// Bumped whenever a change can alter what the cost-volume kernels read or write: profiles/traffic_*.json carries the revision
// its PMC counters were taken with, and bench.py reports `traffic` only when the two agree.
const char *dfe_kernel_revision(void) { return DFE_CV_KERNEL_REV; }

int dfe_ssd_cost_volume_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int kh, int kw,
                            int hWin, int wWin, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, I0 && I1 && out, DFE_E_ARG, "dfe_ssd_cost_volume_f32: NULL tensor");
    DFE_REQUIRE(ctx, C > 0 && kh > 0 && kw > 0 && hWin > 0 && wWin > 0, DFE_E_ARG,
                "dfe_ssd_cost_volume_f32: C=%d k=%dx%d win=%dx%d must be positive", C, kh, kw, hWin, wWin);
    const int Ho = H - kh + 1 - hWin + 1, Wo = W - kw + 1 - wWin + 1;
    DFE_REQUIRE(ctx, Ho > 0 && Wo > 0, DFE_E_SHAPE,
                "dfe_ssd_cost_volume_f32: frame %dx%d too small for kernel %dx%d + window %dx%d", H, W, kh, kw, hWin, wWin);
    return cv_frames_dispatch(ctx, I0, I1, C, H, W, (long long)H * W, kh, kw, hWin, wWin, out);
}

// rows of the output volume that fit one scratch band (ctx->scratch_limit, 16 GiB by default:
// one launch per pair up to 1080p/33x33; 288 GB of HBM make the whole volume the natural unit)
static int band_rows(const dfe_ctx *ctx, int Ho, int Wo, int D, int elem = (int)sizeof(float)) {
    long long row_bytes = (long long)Wo * D * elem;
    long long band = (long long)ctx->scratch_limit / row_bytes;
    if (band < 1) band = 1;
    if (band > Ho) band = Ho;
    // balanced: the same number of bands, rows dealt evenly (band i = rows [i Ho / nb, (i+1) Ho / nb)), so that no short last
    // band is left over -- the fast kernels need at least a tile of rows, and the fp16 pipeline has no other kernel to fall back to
    const long long nb = (Ho + band - 1) / band;
    return (int)((Ho + nb - 1) / nb);
}
static int band_count(int Ho, int band) { return (Ho + band - 1) / band; }

// One pair through build + flow extraction.  Preferred: the fused build (per-chunk minimum + first index, the centre
// cell and the pixel's first 16 cells leave the kernel with the volume, ~210 compact bytes per pixel) + a finalize that
// reads only those; it goes back to the volume only for pixels whose first 16 cells hold fewer than M values above
// the extractOutput threshold.
// Fallback (shapes without a fused instantiation): build, then the full pass dfe_flow_tail.
static int flow_pipeline(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int kh, int kw, int hWin, int wWin,
                         double thr, int64_t *idx, float *best, float *fy, float *fx, float *scores, int64_t *imaxs, int pitch,
                         int pad_t, int pad_l, int scores_padded, const DfePairDepth *pd = nullptr, bool *pd_done = nullptr, float f16_scale = 0.f) {
    // f16_scale != 0: the volume is materialised as fp16 (cost * f16_scale); arg-min, centre and lead cells still come from
    // the fp32 sums in the kernel, so indices and minima are those of the fp32 path.  No extractOutput scores then (its
    // rare fall-back reads the volume).
    const int Ho = H - kh + 1 - hWin + 1, Wo = W - kw + 1 - wWin + 1;
    const int D = hWin * wWin, nch = (D + 63) / 64;
    const long long P = (long long)Ho * Wo;
    const int elem = f16_scale != 0.f ? 2 : 4;
    const int band = band_rows(ctx, Ho, Wo, D, elem);
    const size_t vol_bytes = ((size_t)band * Wo * D * elem + 255) / 256 * 256;
    const size_t part_bytes = ((size_t)nch * P * sizeof(float2) + 255) / 256 * 256;
    const size_t cen_bytes = ((size_t)P * sizeof(float) + 255) / 256 * 256;
    const size_t lead_bytes = ((size_t)P * DFE_LEAD * sizeof(float) + 255) / 256 * 256;
    // per-pixel records of the role-split row-image kernels (one whole 128-B line per pixel instead of entries in the three planes)
    const bool want_rec = kh == kw && kh == 7 && rowimg_writes_records(C, hWin, wWin);
    const size_t rec_bytes = want_rec ? (size_t)dfe_cdiv(Wo, 8) * Ho * DFE_REC * sizeof(float) : 0;   // [tile column][output row][DFE_REC]
    void *scr = nullptr;
    int rc = dfe_scratch(ctx, vol_bytes + part_bytes + cen_bytes + lead_bytes + rec_bytes, &scr);
    if (rc) return rc;
    float *vol = (float *)scr;
    CvFuseArgs fa{};
    fa.part = (float2 *)((char *)scr + vol_bytes);
    fa.centre = (float *)((char *)scr + vol_bytes + part_bytes);
    fa.lead = (float *)((char *)scr + vol_bytes + part_bytes + cen_bytes);
    fa.rec = want_rec ? (float *)((char *)scr + vol_bytes + part_bytes + cen_bytes + lead_bytes) : nullptr;
    fa.rec_rows = Ho;
    fa.Ptot = P;
    {
        const int middle = (wWin + 1) / 2 + wWin * ((hWin + 1) / 2 - 1);   // radial/radial_opticalflow_groundtruth.lua:91
        fa.cmid = (middle - 1) >> 6; fa.lmid = (middle - 1) & 63;
    }
    const int nb = band_count(Ho, band);
    for (int bi = 0; bi < nb; ++bi) {
        const int r0 = (int)((long long)bi * Ho / nb), nr = (int)((long long)(bi + 1) * Ho / nb) - r0;   // (nr <= band)
        const int Hb = nr + kh - 1 + hWin - 1;
        const float *b0 = I0 + (long long)r0 * W, *b1 = I1 + (long long)r0 * W;
        bool fused = false, recs = false;
        int nparts = nch;
        std::unique_ptr<DfeStageScope> match_scope(new DfeStageScope(ctx, DFE_STAGE_MATCH));   // (closed in front of the finalize / tail pass below)
        if (kh == kw && f16_scale != 0.f) {
            fa.row_off = r0;
            rc = cv_frames_dispatch_f16(ctx, b0, b1, C, Hb, W, (long long)H * W, kh, hWin, wWin, f16_scale, vol, &fa, &fused, &recs);
            if (rc) return rc;
            nparts = 2;
            if (!fused) return dfe_fail(ctx, DFE_E_UNSUPPORTED, "no fused fp16 cost-volume kernel for C=%d k=%d win=%dx%d out=%dx%d (band of %d rows)", C, kh, hWin, wWin, Ho, Wo, nr);
        } else if (kh == kw) {
            fa.row_off = r0;
            rc = cv_frames_dispatch_fused(ctx, b0, b1, C, Hb, W, (long long)H * W, kh, hWin, wWin, vol, fa, &fused, &nparts, &recs);
            if (rc) return rc;
        }
        if (fused) {
            match_scope.reset();
            DfeStageScope ex(ctx, DFE_STAGE_EXTRACT);
            // one band: the finalize launch also zeroes the frame border and makes depth (pair step: 2 launches instead of 3)
            const bool frame_mode = pd && nr == Ho;
            rc = dfe_flow_finalize(ctx, fa.part, fa.centre, fa.lead, nparts, P, vol, thr, nr, Wo, hWin, wWin, r0, idx, best, fy, fx, scores,
                                   imaxs, pitch, pad_t, pad_l, scores_padded, frame_mode ? pd : nullptr, (recs && fused) ? fa.rec : nullptr, Ho);
            if (frame_mode && pd_done) *pd_done = true;
        } else {
            rc = cv_frames_dispatch(ctx, b0, b1, C, Hb, W, (long long)H * W, kh, kw, hWin, wWin, vol);
            if (rc) return rc;
            match_scope.reset();
            DfeStageScope ex(ctx, DFE_STAGE_EXTRACT);
            rc = dfe_flow_tail(ctx, vol, nr, Wo, hWin, wWin, thr, r0, idx, best, fy, fx, scores, imaxs, pitch, pad_t, pad_l, scores_padded);
        }
        if (rc) return rc;
    }
    return DFE_OK;
}

int dfe_ssd_flow_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int kh, int kw, int hWin,
                     int wWin, double extract_threshold, int64_t *idx, float *best, float *flow_y, float *flow_x,
                     float *scores, int64_t *imaxs) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, I0 && I1, DFE_E_ARG, "dfe_ssd_flow_f32: NULL frame");
    DFE_REQUIRE(ctx, C > 0 && kh > 0 && kw > 0 && hWin > 0 && wWin > 0, DFE_E_ARG,
                "dfe_ssd_flow_f32: C=%d k=%dx%d win=%dx%d must be positive", C, kh, kw, hWin, wWin);
    const int Ho = H - kh + 1 - hWin + 1, Wo = W - kw + 1 - wWin + 1;
    DFE_REQUIRE(ctx, Ho > 0 && Wo > 0, DFE_E_SHAPE, "dfe_ssd_flow_f32: frame %dx%d too small for kernel %dx%d + window %dx%d",
                H, W, kh, kw, hWin, wWin);
    DFE_REQUIRE(ctx, (scores == nullptr) == (imaxs == nullptr), DFE_E_ARG, "dfe_ssd_flow_f32: scores and imaxs go together");
    return flow_pipeline(ctx, I0, I1, C, H, W, kh, kw, hWin, wWin, extract_threshold, idx, best, flow_y, flow_x, scores, imaxs, Wo, 0,
                         0, 0);
}

int dfe_flow_depth_pair_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int k, int hWin, int wWin,
                            float foe_x, float foe_y, double extract_threshold, float *flow, float *scores, float *depth,
                            float *depth_conf) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, I0 && I1 && flow, DFE_E_ARG, "dfe_flow_depth_pair_f32: NULL tensor");
    DFE_REQUIRE(ctx, C > 0 && k > 0 && hWin > 0 && wWin > 0, DFE_E_ARG, "dfe_flow_depth_pair_f32: C=%d k=%d win=%dx%d", C, k,
                hWin, wWin);
    DFE_REQUIRE(ctx, (depth == nullptr) == (depth_conf == nullptr), DFE_E_ARG,
                "dfe_flow_depth_pair_f32: depth and depth_conf go together");
    const int Ho = H - k + 1 - hWin + 1, Wo = W - k + 1 - wWin + 1;
    DFE_REQUIRE(ctx, Ho > 0 && Wo > 0, DFE_E_SHAPE, "dfe_flow_depth_pair_f32: frame %dx%d too small for kernel %d + window %dx%d",
                H, W, k, hWin, wWin);
    const long long HW = (long long)H * W;
    // centre-paste offsets: opticalflow_model.lua:228-230 floor((hImg-h)/2)
    //   == radial/radial_opticalflow_groundtruth.lua:27-32 floor((hWin-1)/2)+floor((k-1)/2)
    const int pad_t = (H - Ho) / 2, pad_l = (W - Wo) / 2;
    // the pipeline writes every interior pixel of flow / scores; one pass afterwards zeroes the border and makes depth
    const DfePairDepth pd{H, W, foe_x, foe_y, depth, depth_conf};
    bool pd_done = false;
    int rc = flow_pipeline(ctx, I0, I1, C, H, W, k, k, hWin, wWin, extract_threshold, nullptr, nullptr, flow, flow + HW, scores, nullptr,
                           W, pad_t, pad_l, 1, &pd, &pd_done);
    if (rc || pd_done) return rc;
    // several bands, or no fused build for this shape: one pass afterwards zeroes the border and makes depth
    DfeStageScope ex(ctx, DFE_STAGE_EXTRACT);
    return dfe_pair_border_depth(ctx, flow, scores, H, W, pad_t, pad_l, Ho, Wo, foe_x, foe_y, depth, depth_conf);
}

int dfe_ssd_cost_volume_f16(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int kh, int kw, int hWin, int wWin,
                            float scale, void *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, I0 && I1 && out, DFE_E_ARG, "dfe_ssd_cost_volume_f16: NULL tensor");
    DFE_REQUIRE(ctx, C > 0 && kh > 0 && kw > 0 && hWin > 0 && wWin > 0 && scale > 0, DFE_E_ARG,
                "dfe_ssd_cost_volume_f16: C=%d k=%dx%d win=%dx%d scale=%g must be positive", C, kh, kw, hWin, wWin, (double)scale);
    const int Ho = H - kh + 1 - hWin + 1, Wo = W - kw + 1 - wWin + 1;
    DFE_REQUIRE(ctx, Ho > 0 && Wo > 0, DFE_E_SHAPE, "dfe_ssd_cost_volume_f16: frame %dx%d too small for kernel %dx%d + window %dx%d", H, W, kh, kw,
                hWin, wWin);
    if (kh == kw) {
        bool handled = false;
        int rc = cv_frames_dispatch_f16(ctx, I0, I1, C, H, W, (long long)H * W, kh, hWin, wWin, scale, out, nullptr, &handled);
        if (rc != DFE_OK || handled) return rc;
    }
    // any other shape: fp32 bands in the scratch arena, converted
    const int D = hWin * wWin;
    const int band = band_rows(ctx, Ho, Wo, D);
    void *scr = nullptr;
    int rc = dfe_scratch(ctx, (size_t)band * Wo * D * sizeof(float), &scr);
    if (rc) return rc;
    const int nb = band_count(Ho, band);
    for (int bi = 0; bi < nb; ++bi) {
        const int r0 = (int)((long long)bi * Ho / nb), nr = (int)((long long)(bi + 1) * Ho / nb) - r0;
        rc = cv_frames_dispatch(ctx, I0 + (long long)r0 * W, I1 + (long long)r0 * W, C, nr + kh - 1 + hWin - 1, W, (long long)H * W, kh, kw, hWin, wWin,
                                (float *)scr);
        if (rc) return rc;
        const long long n = (long long)nr * Wo * D;
        long long blocks = (n + 255) / 256;
        if (blocks > 256 * 64) blocks = 256 * 64;
        hipLaunchKernelGGL(cv_to_half_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const float *)scr, n, scale,
                           (_Float16 *)out + (long long)r0 * Wo * D);
        DFE_LAUNCH_CHECK(ctx);
    }
    return DFE_OK;
}

int dfe_flow_depth_pair_f16(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int k, int hWin, int wWin, float foe_x,
                            float foe_y, float scale, int64_t *idx, float *best, float *flow, float *depth, float *depth_conf) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, I0 && I1 && flow, DFE_E_ARG, "dfe_flow_depth_pair_f16: NULL tensor");
    DFE_REQUIRE(ctx, C > 0 && k > 0 && hWin > 0 && wWin > 0 && scale > 0, DFE_E_ARG, "dfe_flow_depth_pair_f16: C=%d k=%d win=%dx%d scale=%g", C, k, hWin,
                wWin, (double)scale);
    DFE_REQUIRE(ctx, (depth == nullptr) == (depth_conf == nullptr), DFE_E_ARG, "dfe_flow_depth_pair_f16: depth and depth_conf go together");
    const int Ho = H - k + 1 - hWin + 1, Wo = W - k + 1 - wWin + 1;
    DFE_REQUIRE(ctx, Ho > 0 && Wo > 0, DFE_E_SHAPE, "dfe_flow_depth_pair_f16: frame %dx%d too small for kernel %d + window %dx%d", H, W, k, hWin, wWin);
    const long long HW = (long long)H * W;
    const int pad_t = (H - Ho) / 2, pad_l = (W - Wo) / 2;
    const DfePairDepth pd{H, W, foe_x, foe_y, depth, depth_conf};
    bool pd_done = false;
    // (idx / best are [Ho][Wo], like dfe_ssd_flow_f32's; flow is the centre-pasted [2][H][W])
    int rc = flow_pipeline(ctx, I0, I1, C, H, W, k, k, hWin, wWin, 0.0, idx, best, flow, flow + HW, nullptr, nullptr, W, pad_t, pad_l, 1, &pd, &pd_done,
                           scale);
    if (rc || pd_done) return rc;
    DfeStageScope ex(ctx, DFE_STAGE_EXTRACT);
    return dfe_pair_border_depth(ctx, flow, nullptr, H, W, pad_t, pad_l, Ho, Wo, foe_x, foe_y, depth, depth_conf);
}

int dfe_spatial_matching_f32(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh,
                             int maxw, float *out) {
    DFE_ENTER(ctx);
    return dfe_spatial_matching_dispatch(ctx, in1, in2, K, H1, W1, maxh, maxw, out);
}
}  // extern "C"

int dfe_spatial_matching_dispatch(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh, int maxw, float *out) {
    DFE_REQUIRE(ctx, in1 && in2 && out, DFE_E_ARG, "dfe_spatial_matching_f32: NULL tensor");
    DFE_REQUIRE(ctx, K > 0 && H1 > 0 && W1 > 0 && maxh > 0 && maxw > 0, DFE_E_SHAPE,
                "dfe_spatial_matching_f32: K=%d H1=%d W1=%d maxh=%d maxw=%d must be positive", K, H1, W1, maxh, maxw);
    if (dfe_feat_matching_mfma_takes(ctx, K, H1, W1, maxh, maxw)) {   // opt-in (fm_mfma = 1): the banded GEMM on the matrix cores, costs to 1e-5
        void *nrm = nullptr;
        int rc = dfe_aux_scratch(ctx, dfe_feat_matching_mfma_scratch(H1, W1, maxh, maxw) * sizeof(float), &nrm);
        if (rc) return rc;
        bool handled = false;
        rc = dfe_feat_matching_mfma(ctx, in1, in2, K, H1, W1, maxh, maxw, (float *)nrm, out, nullptr, nullptr, nullptr, &handled);
        if (rc != DFE_OK || handled) return rc;
    }
    {
        bool handled = false;   // lane <-> cell kernel with the feature tile in LDS (feat_matching.hip), bit-identical results
        int rc = dfe_feat_matching_fast(ctx, in1, in2, K, H1, W1, maxh, maxw, out, &handled);
        if (rc != DFE_OK || handled) return rc;
    }
    CvRefArgs a;
    a.a = in1; a.a_plane = (long long)H1 * W1; a.a_pitch = W1; a.a_oy = 0; a.a_ox = 0;
    a.b = in2; a.b_plane = (long long)(H1 + maxh - 1) * (W1 + maxw - 1); a.b_pitch = W1 + maxw - 1;
    a.C = K; a.kh = 1; a.kw = 1; a.Wo = W1; a.hWin = maxh; a.wWin = maxw;
    a.total = (long long)H1 * W1 * maxh * maxw;
    a.out = out;
    return launch_cv_ref(ctx, a);
}

extern "C" {
int dfe_radial_matching_f32(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W, int hWin,
                            float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, in1 && in2 && out, DFE_E_ARG, "dfe_radial_matching_f32: NULL tensor");
    DFE_REQUIRE(ctx, K > 0 && H1 > 0 && W > 0 && hWin > 0, DFE_E_SHAPE,
                "dfe_radial_matching_f32: K=%d H1=%d W=%d hWin=%d must be positive", K, H1, W, hWin);
    return dfe_spatial_matching_f32(ctx, in1, in2, K, H1, W, hWin, 1, out);
}

}  // extern "C"
