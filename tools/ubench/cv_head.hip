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
    int chunk0;        // tiled kernel: first 64-displacement chunk it covers (0 = all; the row-image kernel hands it the tail)
    int stage_off;     // row-span kernel: byte offset of the store-exchange stage inside dynamic LDS
    int stage_len;     // row-span kernel: floats per stage buffer
};

extern __shared__ __attribute__((aligned(16))) char dfe_smem[];

// Tuning-only ablation bits (tools/ablate.sh builds side libraries with -DDFE_ABLATE=n; the product build
// leaves it 0): 1 = no global stores, 2 = no LDS reads inside the row loop, 4 = no SMEM inside the row loop.
#ifndef DFE_ABLATE
#define DFE_ABLATE 0
#endif


// global_store_dword with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset:
// no VALU address arithmetic per store.  Nothing in the kernel reads what it stores, and stores
// need no wait before s_endpgm, so the compiler's vmcnt bookkeeping is not involved.
__device__ __forceinline__ void store_uniform_base(const void *base, unsigned lane_bytes, float v) {
    asm volatile("global_store_dword %0, %1, %2" ::"v"(lane_bytes), "v"(v), "s"(base) : "memory");
}

// All-lanes maximum of EIGHT non-negative ints per lane (one per tile column) in ~30 VALU ops instead of 8 x 6:
// a halving butterfly -- after exchanging with lane^1, lane^2, lane^4 each lane is left with the single column
// (lane & 7), then lane^8, ^16, ^32 finish it.  lane^1 / lane^2 are DPP quad permutes and lane^8 a row rotate
// (all fold into v_max_i32_dpp), lane^4 / ^16 are ds_swizzle and lane^32 a ds_bpermute (LDS crossbar, no LDS memory).
// On return every lane holds the wave maximum of column (lane & 7).
template <int TX> __device__ __forceinline__ int wave_max8_nonneg(const int (&k)[TX], int lane) {
    static_assert(TX == 8, "butterfly is written for 8 columns");
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    int a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int mine = b0 ? k[2 * i + 1] : k[2 * i], other = b0 ? k[2 * i] : k[2 * i + 1];
        a[i] = max(mine, __builtin_amdgcn_update_dpp(0, other, 0xB1, 0xf, 0xf, true));      // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int mine = b1 ? a[2 * i + 1] : a[2 * i], other = b1 ? a[2 * i] : a[2 * i + 1];
        b[i] = max(mine, __builtin_amdgcn_update_dpp(0, other, 0x4E, 0xf, 0xf, true));      // quad_perm [2,3,0,1]
    }
    int c;
    {
        const int mine = b2 ? b[1] : b[0], other = b2 ? b[0] : b[1];
        c = max(mine, __builtin_amdgcn_ds_swizzle(other, 0x101f));                          // lane ^ 4
    }
    c = max(c, __builtin_amdgcn_update_dpp(0, c, 0x128, 0xf, 0xf, true));                   // row_ror:8  == lane ^ 8 pairing
    c = max(c, __builtin_amdgcn_ds_swizzle(c, 0x401f));                                     // lane ^ 16
    c = max(c, __shfl_xor(c, 32));                                                          // lane ^ 32
    return c;
}

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
template <int C, int K, int TX, int NT, int NW, int NQ, bool FUSE>
__global__ __launch_bounds__(NW * 64) void ssd_cv_tiled_kernel(const float *__restrict__ I0, const float *__restrict__ I1,
                                                               float *__restrict__ out, CvTiledArgs p, CvFuseArgs fa) {
    using px_t = typename Px<C>::type;
    constexpr int U = VUnroll<K>::value;
    constexpr int ROWS = U * NQ;
    constexpr int TY = ROWS - (K - 1);
    constexpr int GX = NT * TX;
    constexpr int NE = TX + K - 1;
    px_t *lds = reinterpret_cast<px_t *>(dfe_smem);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0n = blockIdx.x * GX, y0n = blockIdx.y * TY;           // nominal origin
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
        if (FUSE || valid) {
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
            px_t abl_px[(NE + 1) / 2 + 1];
            if (DFE_ABLATE & 2) {
#pragma unroll
                for (int s = 0; s < (NE + 1) / 2 + 1; ++s) abl_px[s] = lp[s];
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
                            if (bb * BS + s < NE) b[s] = (DFE_ABLATE & 2) ? abl_px[s] : lr[bb * BS + s];
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
                        if (!(DFE_ABLATE & 4)) {
#pragma unroll
                            for (int c = 0; c < C; ++c) uload<NE>(an + c * HW, av[c]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    float h[TX];
                    hsum<K, TX>(e, h);
                    const bool emit = (K == 7) ? (q > 0) : (q > 0 || m == K - 1);   // r >= K-1
                    const int y = y0 + r - (K - 1);
                    const bool store_row = emit && y >= y0n && (!(DFE_ABLATE & 1) || hprev[0] == -12345.678f);   // wave-uniform
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
                    if (store_row) {
#pragma unroll
                        for (int x = 0; x < TX; ++x)
                            if (!FUSE || valid) store_uniform_base(orow + (long long)x * D * 4, dbytes, vrow[x]);
                        if constexpr (FUSE) {
                            // Costs are >= 0, so their bit patterns order like integers; key = 0x7f800000 - bits is >= 0 and
                            // larger for smaller costs, idle lanes (key 0) lose.  Butterfly 1 leaves lane L with the chunk
                            // minimum of column L&7; each lane then fetches the 8 column maxima of its 8-lane group, marks
                            // where it attains them, and butterfly 2 (max of 63-lane) yields the FIRST lane that does.
                            int key[TX];
#pragma unroll
                            for (int x = 0; x < TX; ++x) key[x] = valid ? 0x7f800000 - __float_as_int(vrow[x]) : 0;
                            const int wk = wave_max8_nonneg<TX>(key, lane);
                            int cand[TX];
#define DFE_CAND(x) cand[x] = (key[x] == __builtin_amdgcn_ds_swizzle(wk, 0x0018 | ((x) << 5))) ? 64 - lane : 0;
                            // swizzle: lane (L & 0x18) | x of my half holds column x's maximum; 64-lane: earlier lane = larger
                            DFE_CAND(0) DFE_CAND(1) DFE_CAND(2) DFE_CAND(3) DFE_CAND(4) DFE_CAND(5) DFE_CAND(6) DFE_CAND(7)
#undef DFE_CAND
                            const int wl = wave_max8_nonneg<TX>(cand, lane);
                            const long long pg0 = (long long)(fa.row_off + y) * p.Wo + xt;          // pixel of column 0
                            if (lane < TX)
                                fa.part[(long long)chunk * fa.Ptot + pg0 + lane] =
                                    make_float2(__int_as_float(0x7f800000 - wk), __int_as_float(chunk * 64 + 64 - wl));
                            if (chunk == fa.cmid && lane == fa.lmid) {      // the lane that owns the centre cell
#pragma unroll
                                for (int x = 0; x < TX; ++x) fa.centre[pg0 + x] = vrow[x];
                            }
                            if (chunk == 0 && lane < DFE_LEAD) {           // the pixel's first cells, for extractOutput
#pragma unroll
                                for (int x = 0; x < TX; ++x) fa.lead[(pg0 + x) * DFE_LEAD + lane] = valid ? vrow[x] : 0.f;
                            }
                        }
                    }
                }
            }
        }
    }
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
    a.lrows = pl.lrows; a.lcols = pl.lcols; a.pitch = pl.pitch; a.stage_off = 0; a.stage_len = 0;
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
        if (ctx->cv_tyq && nq != ctx->cv_tyq) continue;
        CvTilePlan p4 = plan_cv_tiled<C, K, TX, 4, 8>(nq, Ho, Wo, hWin, wWin, ctx->ncu);
        if (p4.score > bp.score) { bp = p4; best = 40 + nq; }
        CvTilePlan p1 = plan_cv_tiled<C, K, TX, 1, 6>(nq, Ho, Wo, hWin, wWin, ctx->ncu);
        if (p1.score > bp.score) { bp = p1; best = 10 + nq; }
    }
    if (!best) return DFE_OK;   // no tile fits: caller falls back
    *handled = true;
    switch (best) {
        case 42: return launch_cv_tiled_one<C, K, TX, 4, 8, 2>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out);
        case 43: return launch_cv_tiled_one<C, K, TX, 4, 8, 3>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out);
        case 44: return launch_cv_tiled_one<C, K, TX, 4, 8, 4>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out);
        case 45: return launch_cv_tiled_one<C, K, TX, 4, 8, 5>(ctx, bp, I0, I1, H, W, plane, hWin, wWin, out);
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
        if (ctx->cv_tyq && nq != ctx->cv_tyq) continue;
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

int cv_frames_dispatch_fused(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, long long plane, int k, int hWin,
                             int wWin, float *out, const CvFuseArgs &fa, bool *handled) {
    *handled = false;
    if (ctx->cv_mode != 0 && ctx->cv_mode != 2) return DFE_OK;
    if (hWin * wWin < 64) return DFE_OK;   // less than one full chunk: not worth a fused instantiation
    if (C == 3 && k == 7) return launch_cv_tiled_fused<3, 7, 8>(ctx, I0, I1, H, W, plane, hWin, wWin, out, fa, handled);
    if (C == 1 && k == 7) return launch_cv_tiled_fused<1, 7, 8>(ctx, I0, I1, H, W, plane, hWin, wWin, out, fa, handled);
    return DFE_OK;
}

// ------------------------------------------------------------------------------------------
// row-image kernel: 16 adjacent chunks of a tile row leave through an LDS image as whole cache lines
// ------------------------------------------------------------------------------------------
// The tiled kernel above hides its arithmetic completely (160 us of compute at VGA) but is bound by its
// store pattern: 256-B pieces whose neighbours are written later reach 2.9 TB/s, line-aligned contiguous
// bursts 5.7 TB/s.  Here one block = one TX-pixel tile and 16 waves = the first 16
// chunks (1024 displacements) of every pixel, one chunk per thread exactly as above.  Per swept row each wave
// deposits its 64 x TX values into a double-buffered LDS image whose float index is congruent to the
// global float index mod 32; after one barrier the block copies the TX runs of 4 KB out as 128-B-aligned,
// 1-KB-per-wave dwordx4 bursts (head/tail fragments of < 32 floats by dword stores), while the next row is
// already being computed.  Displacement chunks >= 16 (65 of the 1089 cells at 33x33) are written by a
// second launch of the tiled kernel restricted to those chunks.
template <int C, int K, int TX, int NQ>
__global__ __launch_bounds__(1024) void ssd_cv_rowspan_kernel(const float *__restrict__ I0, const float *__restrict__ I1,
                                                              float *__restrict__ out, CvTiledArgs p) {
    using px_t = typename Px<C>::type;
    constexpr int NW = 16;
    constexpr int U = VUnroll<K>::value;
    constexpr int ROWS = U * NQ;
    constexpr int TY = ROWS - (K - 1);
    constexpr int NE = TX + K - 1;
    constexpr int RUN = NW * 64;                 // floats per pixel covered by the block
    constexpr int SL = RUN + 32;                 // stage stride per pixel (multiple of 32 floats)
    px_t *lds = reinterpret_cast<px_t *>(dfe_smem);
    float *stage = reinterpret_cast<float *>(dfe_smem + p.stage_off);   // [2][TX][SL], 128-B aligned

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0n = blockIdx.x * TX, y0n = blockIdx.y * TY;
    const int x0 = min(x0n, p.Wo - TX), y0 = min(y0n, p.Ho - TY);
    const long long HW = p.plane;

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
    const int oy = (p.hWin - 1) >> 1, ox = (p.wWin - 1) >> 1;
    const long long a_base = (long long)(y0 + oy) * p.W + (x0 + ox);

    const int d = wave * 64 + lane;              // < 1024 <= D by the launch condition
    const int dy = d / p.wWin, dx = d - dy * p.wWin;
    const px_t *lp = lds + dy * p.pitch + dx;

    float av[C][NE];
#pragma unroll
    for (int c = 0; c < C; ++c) uload<NE>((cfptr)(I0 + a_base + c * HW), av[c]);

    float ring[U][TX], hprev[TX];
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
            constexpr int NB = (NE > 8) ? 2 : 1;
            constexpr int BS = (NE + NB - 1) / NB;
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                px_t b[BS];
#pragma unroll
                for (int s = 0; s < BS; ++s)
                    if (bb * BS + s < NE) b[s] = (DFE_ABLATE & 16) ? lp[s] : lr[bb * BS + s];
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (C == 3) {
#pragma unroll
                    for (int s = 0; s < BS; ++s)
                        if (bb * BS + s < NE) asm volatile("" ::"v"(b[s].w));   // keep ds_read_b128
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
            if (!(DFE_ABLATE & 4096)) {
                const int rn = min(y0 + oy + r + 1, p.H - 1) - (y0 + oy);
                cfptr an = (cfptr)(I0 + a_base + (long long)rn * p.W);
#pragma unroll
                for (int c = 0; c < C; ++c) uload<NE>(an + c * HW, av[c]);
            }
            __builtin_amdgcn_sched_barrier(0);
            float h[TX], v[TX];
            hsum<K, TX>(e, h);
            if constexpr (K == 7) {
#pragma unroll
                for (int x = 0; x < TX; ++x) {
                    v[x] = (ring[m][x] + ring[(m + 2) % 6][x]) + (ring[(m + 4) % 6][x] + h[x]);
                    ring[(m + 5) % 6][x] = hprev[x] + h[x];
                    hprev[x] = h[x];
                }
            } else {
#pragma unroll
                for (int x = 0; x < TX; ++x) ring[m][x] = h[x];
#pragma unroll
                for (int x = 0; x < TX; ++x) {
                    float t = ring[(m + 1) % K][x];
#pragma unroll
                    for (int i = 2; i <= K; ++i) t += ring[(m + i) % K][x];
                    v[x] = t;
                }
            }
            const bool emit = (K == 7) ? (q > 0) : (q > 0 || m == K - 1);   // r >= K-1
            const int y = y0 + r - (K - 1);
            if (emit && y >= y0n) {                                          // block-uniform
                const long long G0 = ((long long)y * p.Wo + x0) * D;         // global float index of pixel 0's run
                const int a0 = (int)(G0 & 31);
                float *st = stage + (r & 1) * (TX * SL);
                // run x starts at global float G0 + x*D; its image starts at st + x*SL + ((a0 + x*D) & 31)
#pragma unroll
                for (int x = 0; x < TX; ++x) st[x * SL + ((a0 + x * D) & 31) + d] = v[x];
                // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for the previous row's
                // global stores to be acknowledged before every barrier and serialise stores with compute.
                // One barrier per row is enough with two images: a wave re-deposits into image (r&1) only
                // after the barrier of row r+1, which every wave reaches after its copy-out of row r.
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (!(DFE_ABLATE & 1) || hprev[0] == -12345.678f) {
                    // body: TX runs x 256 float4 slots, two per thread; a wave covers 1 KB of one run
#pragma unroll
                    for (int k2 = 0; k2 < (TX * 256) / 1024; ++k2) {
                        const int slot = tid + 1024 * k2;
                        const int x = slot >> 8, j = slot & 255;             // x is wave-uniform
                        const int ax = (a0 + x * D) & 31, head = (32 - ax) & 31;
                        const int nbody4 = ((RUN - head) >> 5) << 3;         // float4 pieces in whole 128-B lines
                        if (j < nbody4) {
                            const float4 *sb = reinterpret_cast<const float4 *>(st + x * SL + ax + head);
                            float4 *gb = reinterpret_cast<float4 *>(out + G0 + (long long)x * D + head);
                            gb[j] = sb[j];
                        }
                    }
                    // fragments: lanes 0..31 of wave x write run x's head, lanes 32..63 its tail (each < 32 floats)
                    if (wave < TX && !(DFE_ABLATE & 32)) {
                        const int x = wave;
                        const int ax = (a0 + x * D) & 31, head = (32 - ax) & 31;
                        const int nbody = ((RUN - head) >> 5) << 5;
                        const int tail0 = head + nbody, ntail = RUN - tail0;
                        const float *sr = st + x * SL + ax;
                        float *gr = out + G0 + (long long)x * D;
                        if (lane < 32) { if (lane < head) gr[lane] = sr[lane]; }
                        else if (lane - 32 < ntail) gr[tail0 + lane - 32] = sr[tail0 + lane - 32];
                    }
                }
            }
        }
    }
}

template <int C, int K, int TX, int NQ>
static int launch_cv_rowspan_one(dfe_ctx *ctx, const float *I0, const float *I1, int H, int W, long long plane, int hWin, int wWin,
                                 float *out, bool *handled) {
    using px_t = typename Px<C>::type;
    constexpr int U = VUnroll<K>::value;
    constexpr int ROWS = U * NQ, TY = ROWS - (K - 1);
    static_assert(TX == 8, "the copy-out deals TX*256 float4 slots to 1024 threads");
    const int Ho = H - K + 1 - hWin + 1, Wo = W - K + 1 - wWin + 1;
    const int D = hWin * wWin;
    *handled = false;
    if (D < 1024 || Ho < TY || Wo < TX) return DFE_OK;   // needs 16 full chunks per pixel
    CvTiledArgs a;
    a.plane = plane;
    a.H = H; a.W = W; a.hWin = hWin; a.wWin = wWin; a.Ho = Ho; a.Wo = Wo;
    a.lrows = ROWS + hWin - 1;
    a.lcols = TX + K - 1 + wWin - 1;
    const int M = Px<C>::bank_mod;
    a.pitch = a.lcols;
    while ((a.pitch - wWin) % M != 0) ++a.pitch;
    size_t tile_bytes = (size_t)a.lrows * a.pitch * sizeof(px_t);
    a.stage_off = (int)((tile_bytes + 127) / 128 * 128);
    a.stage_len = TX * (1024 + 32);
    a.chunk0 = 0;
    size_t lds_bytes = a.stage_off + (size_t)2 * a.stage_len * sizeof(float);
    if (lds_bytes > 160 * 1024) return DFE_OK;
    auto kern = ssd_cv_rowspan_kernel<C, K, TX, NQ>;
    DFE_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    dim3 grid(dfe_cdiv(Wo, TX), dfe_cdiv(Ho, TY));
    {
        DfeProfScope prof(ctx);
        hipLaunchKernelGGL(kern, grid, dim3(1024), lds_bytes, ctx->stream, I0, I1, out, a);
        if (D > 1024 && !(DFE_ABLATE & 2048)) {   // the remaining displacement chunks, scattered-piece pattern (6 % of the bytes at 33x33)
            const int save_mode = ctx->cv_mode, save_tyq = ctx->cv_tyq;
            ctx->cv_chunk0 = 16; ctx->cv_mode = 2; ctx->cv_tyq = 0;
            int rc = cv_frames_dispatch(ctx, I0, I1, C, H, W, plane, K, K, hWin, wWin, out);
            ctx->cv_chunk0 = 0; ctx->cv_mode = save_mode; ctx->cv_tyq = save_tyq;
            if (rc) return rc;
        }
    }
    DFE_LAUNCH_CHECK(ctx);
    ctx->last_kernel = "ssd_cv_rowspan_kernel";
    *handled = true;
    return DFE_OK;
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
// H is the number of frame rows visible to this call (a row band of a taller frame when plane > H*W)
int cv_frames_dispatch(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, long long plane, int kh,
                              int kw, int hWin, int wWin, float *out) {
    const int Ho = H - kh + 1 - hWin + 1, Wo = W - kw + 1 - wWin + 1;
    if ((ctx->cv_mode == 3 || (ctx->cv_mode == 0 && ctx->cv_rowspan)) && kh == kw && C == 3 && kh == 7) {
        bool handled = false;
        int rc;
        switch (ctx->cv_tyq) {
            case 3: rc = launch_cv_rowspan_one<3, 7, 8, 3>(ctx, I0, I1, H, W, plane, hWin, wWin, out, &handled); break;
            case 4: rc = launch_cv_rowspan_one<3, 7, 8, 4>(ctx, I0, I1, H, W, plane, hWin, wWin, out, &handled); break;
            case 6: rc = launch_cv_rowspan_one<3, 7, 8, 6>(ctx, I0, I1, H, W, plane, hWin, wWin, out, &handled); break;
            default: rc = launch_cv_rowspan_one<3, 7, 8, 5>(ctx, I0, I1, H, W, plane, hWin, wWin, out, &handled); break;
        }
        if (rc != DFE_OK || handled) return rc;
        if (ctx->cv_mode == 3)
            return dfe_fail(ctx, DFE_E_UNSUPPORTED, "no row-span cost-volume kernel for C=%d k=%d win=%dx%d out=%dx%d", C, kh, hWin, wWin, Ho, Wo);
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

int dfe_ssd_cost_volume_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int kh, int kw,
                            int hWin, int wWin, float *out) {
    DFE_REQUIRE(ctx, ctx, DFE_E_ARG, "ctx is NULL");
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
static int band_rows(const dfe_ctx *ctx, int Ho, int Wo, int D) {
    long long row_bytes = (long long)Wo * D * sizeof(float);
    long long band = (long long)ctx->scratch_limit / row_bytes;
    if (band < 1) band = 1;
    if (band > Ho) band = Ho;
    return (int)band;
}

// One pair through build + flow extraction.  Preferred: the fused build (per-chunk minimum + first index, the centre
// cell and the pixel's first 16 cells leave the kernel with the volume, ~210 compact bytes per pixel) + a finalize that
// reads only those; it goes back to the volume only for pixels whose first 16 cells hold fewer than M values above
// the extractOutput threshold.
// Fallback (shapes without a fused instantiation): build, then the full pass dfe_flow_tail.
static int flow_pipeline(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int kh, int kw, int hWin, int wWin,
                         double thr, int64_t *idx, float *best, float *fy, float *fx, float *scores, int64_t *imaxs, int pitch,
                         int pad_t, int pad_l, int scores_padded) {
    const int Ho = H - kh + 1 - hWin + 1, Wo = W - kw + 1 - wWin + 1;
    const int D = hWin * wWin, nch = (D + 63) / 64;
    const long long P = (long long)Ho * Wo;
    const int band = band_rows(ctx, Ho, Wo, D);
    const size_t vol_bytes = ((size_t)band * Wo * D * sizeof(float) + 255) / 256 * 256;
    const size_t part_bytes = ((size_t)nch * P * sizeof(float2) + 255) / 256 * 256;
    const size_t cen_bytes = ((size_t)P * sizeof(float) + 255) / 256 * 256;
    const size_t lead_bytes = ((size_t)P * DFE_LEAD * sizeof(float) + 255) / 256 * 256;
    void *scr = nullptr;
    int rc = dfe_scratch(ctx, vol_bytes + part_bytes + cen_bytes + lead_bytes, &scr);
    if (rc) return rc;
    float *vol = (float *)scr;
    CvFuseArgs fa{};
    fa.part = (float2 *)((char *)scr + vol_bytes);
    fa.centre = (float *)((char *)scr + vol_bytes + part_bytes);
    fa.lead = (float *)((char *)scr + vol_bytes + part_bytes + cen_bytes);
    fa.Ptot = P;
    {
        const int middle = (wWin + 1) / 2 + wWin * ((hWin + 1) / 2 - 1);   // radial/radial_opticalflow_groundtruth.lua:91
        fa.cmid = (middle - 1) >> 6; fa.lmid = (middle - 1) & 63;
    }
    for (int r0 = 0; r0 < Ho; r0 += band) {
        const int nr = (r0 + band <= Ho) ? band : Ho - r0;
        const int Hb = nr + kh - 1 + hWin - 1;
        const float *b0 = I0 + (long long)r0 * W, *b1 = I1 + (long long)r0 * W;
        bool fused = false;
        if (kh == kw) {
            fa.row_off = r0;
            rc = cv_frames_dispatch_fused(ctx, b0, b1, C, Hb, W, (long long)H * W, kh, hWin, wWin, vol, fa, &fused);
            if (rc) return rc;
        }
        if (fused) {
            rc = dfe_flow_finalize(ctx, fa.part, fa.centre, fa.lead, nch, P, vol, thr, nr, Wo, hWin, wWin, r0, idx, best, fy, fx, scores,
                                   imaxs, pitch, pad_t, pad_l, scores_padded);
        } else {
            rc = cv_frames_dispatch(ctx, b0, b1, C, Hb, W, (long long)H * W, kh, kw, hWin, wWin, vol);
            if (rc) return rc;
            rc = dfe_flow_tail(ctx, vol, nr, Wo, hWin, wWin, thr, r0, idx, best, fy, fx, scores, imaxs, pitch, pad_t, pad_l, scores_padded);
        }
        if (rc) return rc;
    }
    return DFE_OK;
}

int dfe_ssd_flow_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int kh, int kw, int hWin,
                     int wWin, double extract_threshold, int64_t *idx, float *best, float *flow_y, float *flow_x,
                     float *scores, int64_t *imaxs) {
    DFE_REQUIRE(ctx, ctx, DFE_E_ARG, "ctx is NULL");
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
    DFE_REQUIRE(ctx, ctx, DFE_E_ARG, "ctx is NULL");
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
    DFE_HIP(ctx, hipMemsetAsync(flow, 0, 2 * HW * sizeof(float), ctx->stream));
    if (scores) DFE_HIP(ctx, hipMemsetAsync(scores, 0, HW * sizeof(float), ctx->stream));
    int rc = flow_pipeline(ctx, I0, I1, C, H, W, k, k, hWin, wWin, extract_threshold, nullptr, nullptr, flow, flow + HW, scores, nullptr,
                           W, pad_t, pad_l, 1);
    if (rc) return rc;
    if (depth) {
        rc = dfe_flow_to_depth_cartesian(ctx, flow, H, W, foe_x, foe_y, 0, depth, depth_conf);
        if (rc) return rc;
    }
    return DFE_OK;
}

int dfe_spatial_matching_f32(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh,
                             int maxw, float *out) {
    DFE_REQUIRE(ctx, ctx, DFE_E_ARG, "ctx is NULL");
    DFE_REQUIRE(ctx, in1 && in2 && out, DFE_E_ARG, "dfe_spatial_matching_f32: NULL tensor");
    DFE_REQUIRE(ctx, K > 0 && H1 > 0 && W1 > 0 && maxh > 0 && maxw > 0, DFE_E_SHAPE,
                "dfe_spatial_matching_f32: K=%d H1=%d W1=%d maxh=%d maxw=%d must be positive", K, H1, W1, maxh, maxw);
    CvRefArgs a;
    a.a = in1; a.a_plane = (long long)H1 * W1; a.a_pitch = W1; a.a_oy = 0; a.a_ox = 0;
    a.b = in2; a.b_plane = (long long)(H1 + maxh - 1) * (W1 + maxw - 1); a.b_pitch = W1 + maxw - 1;
    a.C = K; a.kh = 1; a.kw = 1; a.Wo = W1; a.hWin = maxh; a.wWin = maxw;
    a.total = (long long)H1 * W1 * maxh * maxw;
    a.out = out;
    return launch_cv_ref(ctx, a);
}

int dfe_radial_matching_f32(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W, int hWin,
                            float *out) {
    DFE_REQUIRE(ctx, ctx, DFE_E_ARG, "ctx is NULL");
    DFE_REQUIRE(ctx, in1 && in2 && out, DFE_E_ARG, "dfe_radial_matching_f32: NULL tensor");
    DFE_REQUIRE(ctx, K > 0 && H1 > 0 && W > 0 && hWin > 0, DFE_E_SHAPE,
                "dfe_radial_matching_f32: K=%d H1=%d W=%d hWin=%d must be positive", K, H1, W, hWin);
    return dfe_spatial_matching_f32(ctx, in1, in2, K, H1, W, hWin, 1, out);
}

}  // extern "C"
