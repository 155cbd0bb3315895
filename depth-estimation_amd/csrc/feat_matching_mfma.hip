// feat_matching_mfma.hip -- nn.SpatialMatching(17, 17) / (16, 16) on K-plane feature maps as a BANDED GEMM on the matrix cores (opt-in:
// dfe_set_option(ctx, "fm_mfma", 1); version2/network.lua:30, opticalflow_model.lua:93, tests/time_matching.lua:18).
//   out[y][x][dy][dx] = sum_k (a_k - b_k)^2 = |a|^2 + |b|^2 - 2 a.b,     a = in1[:][y][x],  b = in2[:][y + dy][x + dx]
// The exact kernels (feat_matching_flat.hip) sum the squared differences plane by plane on the vector ALUs -- 3 K lane-operations per
// output, 0.205 ms for version2's VGA pair, VALU-issue-bound.  Here the K-contraction a.b is what v_mfma_f32_16x16x4_f32 does: for one
// output row y, 16 pixels x0 .. x0+15 and one window row dy, the products against the 32 columns x0 .. x0+31 of in2's row y + dy are
// two 16 x 16 tiles; the window's 17 cells of a pixel are the band dx = x' - x in [0, 17) of them (272 of 512 products used: the
// price of a dense tile on a banded problem).  K = 32: 16 MFMAs per (16 pixels, dy), 4.6 M per VGA pair = 60 us of matrix-core time
// against the 120 us the vector form needs at its best.
// NUMERICS: the cost is ONE fmaf chain of the MFMA: |a|^2 * 1 and 1 * |b|^2 first (the norms, summed in fp32 by fmm_norm_kernel or by the
// convolution's epilogue, ride in the GEMM as an extra k-step), then (-2 a_k) b_k in k order.  It differs from the exact sum of squared differences by cancellation: about 1e-7 (|a|^2 + |b|^2) absolute,
// i.e. the relative error of a SMALL cost (a good match) is larger than that of the exact kernel.  Tolerance as tested
// (tests/test_gpu_matcher_full.py): |c - exact| <= 1e-5 |exact| + 1e-6 max|exact|; arg-min equal except where the two best exact
// costs lie within that band.  The exact kernels stay the default.
//   block = 8 waves = 8 output rows x 16 pixels, one block per CU; wave w owns row w: 17 x 2 accumulator tiles (136 registers).  (Tried:
//   two blocks of 4 waves per CU so that one block's MFMAs run beside the other's epilogue -- 165 against 137 us: twice the in2 rows
//   staged per output row.)  Planes are staged 8
//   at a time -- in2's 24 rows x 32 columns and in1's 8 rows x 16 -- by LDS-DMA into a double-buffered LDS tile (the requests of
//   stage s+1 in flight behind the MFMAs of stage s: no staging registers next to the accumulators); per MFMA one ds_read_b32 of the
//   B operand at an immediate offset.
#include "dfe_internal.h"

#ifndef DFE_FMM_PIPE
#define DFE_FMM_PIPE 1
#endif
#ifndef DFE_FMM_KC16
#define DFE_FMM_KC16 1
#endif
namespace {

template <int I, int N, class F> __device__ __forceinline__ void fmm_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        fmm_static_for<I + 1, N>(f);
    }
}

typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int FMM_R = 8;     // output rows per block (= waves)
constexpr int FMM_KC = 8;    // planes per stage (two MFMA k-steps); the arg-min form takes 16 where K % 16 == 0
constexpr int FMM_T = FMM_R * 64;

struct FmmArgs {
    const float *in1, *in2, *na, *nb;   // feature maps and their per-pixel squared norms
    float *out;                          // [H1][W1][MH][MW], or NULL
    long long *idx;                      // ARGMIN outputs (each may be NULL)
    float *xflow, *yflow;
    int K, H1, W1, H2, W2;
    int gx, ntiles;                      // 16-pixel groups per output row; tiles (8 rows x 16 pixels)
    int lWin, tWin;
};

// squared norm over the planes: out[p] = fma chain over k of in[k][p]^2
__global__ __launch_bounds__(256) void fmm_norm_kernel(const float *__restrict__ in, int K, long long P, float *__restrict__ out) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) {
            const float v = in[(long long)k * P + p];
            s = fmaf(v, v, s);
        }
        out[p] = s;
    }
}

typedef __attribute__((address_space(3))) float fmm_lds_f;
// one LDS-DMA request: lane l fetches the float at sbase + voff (bytes; its own offset) into LDS at lds_dst + 4 l (M0 is the compiler's:
// saved and restored around the instruction -- cdna_hip_programming.md, LDS-DMA recipe)
__device__ __forceinline__ void fmm_glds4(unsigned voff, const void *sbase, const float *lds_dst) {
    const unsigned la = (unsigned)(size_t)(const fmm_lds_f *)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(la) : "memory");
}

// ... 16 bytes per lane: LDS at lds_dst + 16 l
__device__ __forceinline__ void fmm_glds16(unsigned voff, const void *sbase, const float *lds_dst) {
    const unsigned la = (unsigned)(size_t)(const fmm_lds_f *)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(la) : "memory");
}

template <int MH, int MW, bool ARGMIN, int KC = FMM_KC>
__global__ __launch_bounds__(FMM_T) void fmm_kernel(FmmArgs p) {
    static_assert(MW == 16 || MW == 17, "the band is one tile and a triangle of the next");
    constexpr int R = FMM_R;
    constexpr int BROWS = (R + MH - 1 + 3) & ~3;          // in2 rows of a tile (20: rounded up so that requests of 2 / 4 / 8 rows tile it)
    constexpr int BPS = BROWS * 32 + 16;                  // floats per staged in2 plane (+16: lanes kq and kq + 1 on different banks)
    constexpr int APS = R * 16;                           // ... per staged in1 plane (one A read per 2 MH MFMAs: its 2-way conflict does not matter)
    constexpr int NRB = KC * BROWS / 2, NRA = KC * R / 4; // 4-byte requests of 64 floats per stage: in2 row pairs, in1 row quadruples
    constexpr int BUF = KC * BPS + KC * APS;              // floats per stage buffer
    extern __shared__ __attribute__((aligned(16))) float fmm_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kq = lane >> 4;
    const unsigned plane1 = (unsigned)(p.H1 * p.W1), plane2 = (unsigned)(p.H2 * p.W2);
    const int nstages = (p.K + KC - 1) / KC;
    // the virtual block index puts the blocks of an XCD (linear ids b, b + 8, ...) on consecutive tiles: neighbours share in2 rows in L2
    const int nbx = gridDim.x;
    int vb = blockIdx.x;
    if (!(nbx & 7)) vb = (int)(blockIdx.x & 7) * (nbx >> 3) + (int)(blockIdx.x >> 3);
    // Staging by LDS-DMA, no registers and no LDS stores.  Request n = wave + 8 jj of a stage is wave-uniform; a lane's source row is clamped
    // at the frame's edge (clamped values feed masked outputs only).
    //   interior tiles (every tile of a frame whose width is a multiple of 16): 16 bytes per lane -- a request is 8 rows x 32 columns of an
    //     in2 plane or two whole in1 planes' 8 x 16: 28 requests per stage (first version: 4-byte requests everywhere, 112 per stage:
    //     0.22 ms per VGA pair -- the DMA issue rate, not the MFMAs, set the pace);
    //   tiles of a ragged last column (a 16-byte piece cannot be clamped inside) and K % 8 != 0: 4 bytes per lane, planes past K zero-filled
    //     by ordinary LDS stores.
    // Stage 0 of a tile also brings the squared norms of its in2 pixels ([24][32], three more requests) -- and it is requested during
    // the LAST stage of the tile before (first version: at the top of the tile, its round trip to memory exposed once per tile).
    static_assert(R == 8 && KC % 2 == 0 && BROWS % 4 == 0, "x4 requests: 8-row pieces of in2 planes (the last one may be partial), two in1 planes of 8 x 16");
    constexpr int NXP = (BROWS + 7) / 8;                  // 16-byte requests per in2 plane
    constexpr int NXB = KC * NXP, NXA = KC / 2, NXN = NXP;
    constexpr int N1N = BROWS / 2;
    auto issue = [&](int t, int sg, float *buf, float *nbuf) {
        const int ty = t / p.gx, tx = t - ty * p.gx;
        const int y0 = ty * R, x0 = tx * 16;
        const bool x4 = x0 + 32 <= p.W2 && x0 + 16 <= p.W1 && p.K % KC == 0;   // (block-uniform)
        const int nextra = sg == 0 ? 1 : 0;
        if (x4) {
            for (int n = wave; n < NXB + NXA + nextra * NXN; n += R) {          // (wave-uniform)
                if (n < NXB) {
                    const int kc = n / NXP, r8 = n - kc * NXP;
                    const int row = 8 * r8 + (lane >> 3), col = 4 * (lane & 7);
                    const unsigned off = (unsigned)(sg * KC + kc) * plane2 + (unsigned)min(y0 + row, p.H2 - 1) * (unsigned)p.W2 + (unsigned)(x0 + col);
                    if (row < BROWS) fmm_glds16(4u * off, p.in2, buf + kc * BPS + r8 * 256);
                } else if (n < NXB + NXA) {
                    const int kc = 2 * (n - NXB) + (lane >> 5);
                    const int row = (lane >> 2) & 7, col = 4 * (lane & 3);
                    const unsigned off = (unsigned)(sg * KC + kc) * plane1 + (unsigned)min(y0 + row, p.H1 - 1) * (unsigned)p.W1 + (unsigned)(x0 + col);
                    fmm_glds16(4u * off, p.in1, buf + KC * BPS + 2 * (n - NXB) * APS);
                } else {
                    const int r8 = n - NXB - NXA;
                    const int row = 8 * r8 + (lane >> 3), col = 4 * (lane & 7);
                    if (row < BROWS) fmm_glds16(4u * ((unsigned)min(y0 + row, p.H2 - 1) * (unsigned)p.W2 + (unsigned)(x0 + col)), p.nb, nbuf + r8 * 256);
                }
            }
        } else {
            for (int n = wave; n < NRB + NRA + nextra * N1N; n += R) {
                if (n < NRB) {
                    const int kc = n / (BROWS / 2), rp = n - kc * (BROWS / 2);
                    const int row = 2 * rp + (lane >> 5), col = lane & 31, k = sg * KC + kc;
                    float *dst = buf + kc * BPS + rp * 64;
                    if (k >= p.K) { dst[lane] = 0.f; continue; }
                    fmm_glds4(4u * ((unsigned)k * plane2 + (unsigned)min(y0 + row, p.H2 - 1) * (unsigned)p.W2 + (unsigned)min(x0 + col, p.W2 - 1)), p.in2, dst);
                } else if (n < NRB + NRA) {
                    const int n2 = n - NRB, kc = n2 / (R / 4), hf = n2 - kc * (R / 4);
                    const int row = 4 * hf + (lane >> 4), col = lane & 15, k = sg * KC + kc;
                    float *dst = buf + KC * BPS + kc * APS + hf * 64;
                    if (k >= p.K) { dst[lane] = 0.f; continue; }
                    fmm_glds4(4u * ((unsigned)k * plane1 + (unsigned)min(y0 + row, p.H1 - 1) * (unsigned)p.W1 + (unsigned)min(x0 + col, p.W1 - 1)), p.in1, dst);
                } else {
                    const int rp = n - NRB - NRA;
                    const int row = 2 * rp + (lane >> 5), col = lane & 31;
                    fmm_glds4(4u * ((unsigned)min(y0 + row, p.H2 - 1) * (unsigned)p.W2 + (unsigned)min(x0 + col, p.W2 - 1)), p.nb, nbuf + rp * 64);
                }
            }
        }
    };
    float *nbt0 = fmm_smem + 2 * BUF;                                 // [2][BROWS][32]: the norms of this tile's and of the next tile's in2 pixels
    int gs = 0, tp = 0;                                               // stage / tile counters: buffer = parity
    if (vb < p.ntiles) issue(vb, 0, fmm_smem, nbt0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = vb; t < p.ntiles; t += nbx, ++tp) {
        const int ty = t / p.gx, tx = t - ty * p.gx;
        const int y0 = ty * R, x0 = tx * 16;
        const float *nbt = nbt0 + (tp & 1) * (BROWS * 32);
        const int y = y0 + wave;
        const int q = kq;
        // The norms ride in the GEMM as one more k-step (k = K: |a|^2 against 1, k = K + 1: 1 against |b|^2, two zero taps), and the features of
        // in1 enter scaled by -2 (exact): the accumulator IS the cost |a|^2 + |b|^2 - 2 a.b, no per-cell arithmetic behind the MFMAs.  It
        // starts at 0 inside the window's band and at +inf outside it (tile 0: dx = j - i >= 0, tile 1: dx = 16 + j - i < MW), so cells
        // outside the band never win a minimum and need no mask either.
        const float na_i = p.na[(long long)min(y, p.H1 - 1) * p.W1 + min(x0 + j, p.W1 - 1)];   // (A operand of the extra step: pixel i = lane & 15)
        f4v acc[MH][2];
        f4v pen[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            pen[0][r] = j - i >= 0 ? 0.f : __int_as_float(0x7f800000);
            pen[1][r] = 16 + j - i < MW ? 0.f : __int_as_float(0x7f800000);
        }
        // ---- the norms' k-step FIRST, with the band penalties as its C operand: lane (i or j, kq): A = (|a_i|^2, 1, 0, 0)[kq],
        // B = (1, |b|^2, 0, 0)[kq]; the accumulators start as pen + |a|^2 + |b|^2 without 136 register copies (next to f32 MFMAs a vector
        // instruction is not free).  (The norms of this tile's in2 pixels came with its stage 0, complete behind the last barrier.)
        {
            const float *nbw = nbt + wave * 32 + j;
            const float m0 = kq == 0 ? 1.f : 0.f, m1 = kq == 1 ? 1.f : 0.f;
            const float a = kq == 0 ? na_i : m1;
            float b[MH][2];
#pragma unroll
            for (int dy = 0; dy < MH; ++dy) { b[dy][0] = fmaf(m1, nbw[dy * 32], m0); b[dy][1] = fmaf(m1, nbw[dy * 32 + 16], m0); }
#pragma unroll
            for (int dy = 0; dy < MH; ++dy)
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) acc[dy][tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[dy][tl], pen[tl], 0, 0, 0);
        }
        for (int sg = 0; sg < nstages; ++sg, ++gs) {
            {   // the next stage -- of this tile, or stage 0 of the next one -- in flight behind this stage's MFMAs
                const bool same = sg + 1 < nstages;
                const int it = same ? t : t + nbx, isg = same ? sg + 1 : 0;
                if (it < p.ntiles) issue(it, isg, fmm_smem + ((gs + 1) & 1) * BUF, nbt0 + ((tp + 1) & 1) * (BROWS * 32));
            }
            const float *sb = fmm_smem + (gs & 1) * BUF;
            const float *Bw = sb + kq * BPS + wave * 32 + j;          // lane (j, kq): in2 plane kq, tile row wave + dy, column 16 tile + j
            const float *Aw = sb + KC * BPS + kq * APS + wave * 16 + j;
            if constexpr (ARGMIN && DFE_FMM_PIPE) {
                // The operands of k-step ks + 1 are requested BETWEEN the MFMAs of k-step ks (one read behind every MFMA: the scheduler is told
                // so), in a second register set -- the arg-min form has the registers (no volume addressing).  With all reads in front of a
                // k-step's MFMAs the two waves of a SIMD, which leave every barrier together, both sat in their reads at once and the matrix
                // pipe idled a quarter of the time; only the first k-step of a stage is still exposed (its data arrive behind a barrier).
                float ac = -2.f * Aw[0];
                float bc[MH][2];
#pragma unroll
                for (int dy = 0; dy < MH; ++dy) { bc[dy][0] = Bw[dy * 32]; bc[dy][1] = Bw[dy * 32 + 16]; }
                __builtin_amdgcn_sched_barrier(0);                    // (the first k-step's reads stay in front of its MFMAs)
                fmm_static_for<0, KC / 4>([&](auto ksc) {
                    constexpr int ks = decltype(ksc)::value;
                    float an = 0.f;
                    float bn[MH][2];
                    if constexpr (ks + 1 < KC / 4) {
                        an = -2.f * Aw[(ks + 1) * 4 * APS];
#pragma unroll
                        for (int dy = 0; dy < MH; ++dy) { bn[dy][0] = Bw[(ks + 1) * 4 * BPS + dy * 32]; bn[dy][1] = Bw[(ks + 1) * 4 * BPS + dy * 32 + 16]; }
                    }
#pragma unroll
                    for (int dy = 0; dy < MH; ++dy)
#pragma unroll
                        for (int tl = 0; tl < 2; ++tl) acc[dy][tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac, bc[dy][tl], acc[dy][tl], 0, 0, 0);
                    if constexpr (ks + 1 < KC / 4) {
#pragma unroll
                        for (int n = 0; n < 2 * MH; ++n) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read of the next k-step
                        }
                        ac = an;
#pragma unroll
                        for (int dy = 0; dy < MH; ++dy) { bc[dy][0] = bn[dy][0]; bc[dy][1] = bn[dy][1]; }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else {
#pragma unroll
            for (int ks = 0; ks < KC / 4; ++ks) {
                // all 2 MH + 1 operands of the k-step are requested before its first MFMA
                const float a = -2.f * Aw[ks * 4 * APS];
                float b[MH][2];
#pragma unroll
                for (int dy = 0; dy < MH; ++dy) { b[dy][0] = Bw[ks * 4 * BPS + dy * 32]; b[dy][1] = Bw[ks * 4 * BPS + dy * 32 + 16]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dy = 0; dy < MH; ++dy)
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl) acc[dy][tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[dy][tl], acc[dy][tl], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my requests of the next stage have landed; behind the barrier everyone's have
            __syncthreads();
        }
        // ---- epilogue: lane (q = kq, j) holds the costs of pixels i = 4 q + r against column j of every (dy, tile)
        if constexpr (ARGMIN) {
            // pass 1: the pixel's minimum -- per lane over its 2 MH cells, then over the 16 lanes of the row (one v_min per cell; the
            // first version tracked (minimum, cell) with a compare and two selects per cell: 1100 vector instructions per wave and tile,
            // as long as the MFMAs and not overlapped with them)
            float mn[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) mn[r] = acc[0][0][r];
#pragma unroll
            for (int dy = 0; dy < MH; ++dy)
#pragma unroll
                for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mn[r] = fminf(mn[r], acc[dy][tl][r]);   // (the compiler's mix of v_min / v_min3 is the fastest of three forms tried)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#define FMM_STEP(ctrl) mn[r] = fminf(mn[r], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mn[r]), ctrl, 0xf, 0xf, false)));
                FMM_STEP(0x128) FMM_STEP(0x124) FMM_STEP(0x4E) FMM_STEP(0xB1)
#undef FMM_STEP
            }
            // pass 2: the first cell (window order) that attains it: per lane the first (dy, tile) -- walked backwards, the last write wins --
            // then the smallest window index over the row's lanes
            int code[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) code[r] = 0x7fff;
#pragma unroll
            for (int dy = MH - 1; dy >= 0; --dy)
#pragma unroll
                for (int tl = 1; tl >= 0; --tl)
#pragma unroll
                    for (int r = 0; r < 4; ++r) code[r] = acc[dy][tl][r] == mn[r] ? 2 * dy + tl : code[r];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * q + r;
                int bi = code[r] == 0x7fff ? 0x7fffffff : (code[r] >> 1) * MW + ((code[r] & 1) ? 16 + j - i : j - i);
#define FMM_STEP(ctrl) bi = min(bi, __builtin_amdgcn_update_dpp(0, bi, ctrl, 0xf, 0xf, false));
                FMM_STEP(0x128) FMM_STEP(0x124) FMM_STEP(0x4E) FMM_STEP(0xB1)
#undef FMM_STEP
                if (j == 0 && y < p.H1 && x0 + i < p.W1) {
                    if (bi == 0x7fffffff) bi = 0;
                    const long long px = (long long)y * p.W1 + x0 + i;
                    const int fy = bi / MW;
                    if (p.idx) p.idx[px] = (long long)bi + 1;
                    if (p.yflow) p.yflow[px] = (float)(fy - p.tWin);
                    if (p.xflow) p.xflow[px] = (float)(bi - fy * MW - p.lWin);
                }
            }
        } else {
            // the volume, cell by cell (64-byte runs per pixel and window row: this form exists for the tolerance tests and for callers
            // that want the matrix-core costs themselves; the fast path of the one-call models is the arg-min form)
            if (y < p.H1) {
#pragma unroll
                for (int dy = 0; dy < MH; ++dy)
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 4 * q + r, dx = tl ? 16 + j - i : j - i;
                            if (dx >= 0 && dx < MW && x0 + i < p.W1) p.out[(((long long)y * p.W1 + x0 + i) * MH + dy) * MW + dx] = acc[dy][tl][r];
                        }
            }
        }
        // (no barrier here: what the epilogue reads -- this tile's norms -- is overwritten two tiles on, behind the barriers of the next tile's stages)
    }
}

}  // namespace

// floats of scratch the launcher needs for the two norm planes
size_t dfe_feat_matching_mfma_scratch(int H1, int W1, int maxh, int maxw) { return (size_t)H1 * W1 + (size_t)(H1 + maxh - 1) * (W1 + maxw - 1); }

bool dfe_feat_matching_mfma_takes(const dfe_ctx *ctx, int K, int H1, int W1, int maxh, int maxw) {
    if (ctx->opt[DFE_OPT_FM_MFMA] <= 0 || ctx->cv_mode == 1) return false;
    if (!((maxh == 17 && maxw == 17) || (maxh == 16 && maxw == 16))) return false;
    if (K < 1 || K > 256 || H1 < 1 || W1 < 1) return false;
    return (long long)K * (H1 + maxh - 1) * (W1 + maxw - 1) < (1ll << 31);
}

// out != NULL: the volume; else the first-minimum decode (idx / xflow / yflow).  norms: dfe_feat_matching_mfma_scratch floats.
int dfe_feat_matching_mfma(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh, int maxw, float *norms, float *out, long long *idx,
                           float *xflow, float *yflow, bool *handled, bool norms_ready) {
    *handled = false;
    if (!dfe_feat_matching_mfma_takes(ctx, K, H1, W1, maxh, maxw) || !norms) return DFE_OK;
    const int H2 = H1 + maxh - 1, W2 = W1 + maxw - 1;
    float *na = norms, *nb = norms + (size_t)H1 * W1;
    const long long P1 = (long long)H1 * W1, P2 = (long long)H2 * W2;
    if (!norms_ready) {
        hipLaunchKernelGGL(fmm_norm_kernel, dim3((unsigned)std::min<long long>((P1 + 255) / 256, 4096)), dim3(256), 0, ctx->stream, in1, K, P1, na);
        hipLaunchKernelGGL(fmm_norm_kernel, dim3((unsigned)std::min<long long>((P2 + 255) / 256, 4096)), dim3(256), 0, ctx->stream, in2, K, P2, nb);
        DFE_LAUNCH_CHECK(ctx);
    }
    FmmArgs a{};
    a.in1 = in1; a.in2 = in2; a.na = na; a.nb = nb; a.out = out; a.idx = idx; a.xflow = xflow; a.yflow = yflow;
    a.K = K; a.H1 = H1; a.W1 = W1; a.H2 = H2; a.W2 = W2;
    a.gx = dfe_cdiv(W1, 16); a.ntiles = a.gx * dfe_cdiv(H1, FMM_R);
    a.lWin = (maxw + 1) / 2 - 1; a.tWin = (maxh + 1) / 2 - 1;
    const int BROWS = (FMM_R + maxh - 1 + 3) & ~3;
    // the arg-min form stages 16 planes at a time where K allows it: three of four k-steps then have their operands requested behind MFMAs
    const bool kc16 = DFE_FMM_KC16 && !out && K % 16 == 0;
    const int KC = kc16 ? 16 : FMM_KC;
    const size_t lds = ((size_t)2 * (KC * (BROWS * 32 + 16) + KC * (FMM_R * 16)) + (size_t)2 * BROWS * 32) * sizeof(float);
    void (*kern)(FmmArgs) = out      ? (maxh == 17 ? fmm_kernel<17, 17, false> : fmm_kernel<16, 16, false>)
                            : kc16 ? (maxh == 17 ? fmm_kernel<17, 17, true, 16> : fmm_kernel<16, 16, true, 16>)
                                   : (maxh == 17 ? fmm_kernel<17, 17, true> : fmm_kernel<16, 16, true>);
    DFE_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nblk = a.ntiles < ctx->ncu ? a.ntiles : ctx->ncu;
    {
        DfeProfScope prof(ctx);
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(FMM_T), lds, ctx->stream, a);
    }
    DFE_LAUNCH_CHECK(ctx);
    ctx->last_kernel = out ? "fmm_kernel" : "fmm_kernel+argmin";
    *handled = true;
    return DFE_OK;
}
