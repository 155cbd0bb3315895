// conv_mfma.hip -- next-row N1: nn.SpatialConvolution as an implicit GEMM on the matrix cores.
//   replaces: the convolutions of getFilter (opticalflow_model.lua:45-79: {3,5,5,4},{4,5,5,4},{4,5,5,10} in
//   tests/time_matching.lua:13), of version2/network.lua:12-19 (3 -> 32 planes, 17 x 17) and of
//   radial/radial_opticalflow_network.lua:6-30, when the caller asks for the fast kernel (dfe_set_convolution_kernel).
// GEMM view: M = output pixels, N = output planes (tiles of 16), K = (input plane, ky, kx).  v_mfma_f32_16x16x4_f32: f32
// operands, f32 accumulate -- on gfx950 bitwise an fmaf chain over k in order (MI355X_MICROARCH.md), so
//     out = fma(w[K-1], in[K-1], ... fma(w[1], in[1], fma(w[0], in[0], bias)))        with k = (i, u, v), v fastest,
// i.e. the reference order with fused instead of separately rounded multiply-adds: tolerance 1e-5 relative to sum |terms|
// against nn.SpatialConvolution's loop, bit-exact against the oracle's fmaf variant.  The matrix cores run f32 at the vector
// FMA rate, but one instruction replaces 4 x 16 x 16 multiply-adds: the direct kernel needs two VALU instructions per
// multiply-add (no fusing: bit-exactness with the CPU loop) and one accumulator register per output.
//   block = 4 waves = 4 output rows x 64 columns; per input plane: the (kH + 3) x (64 + kWp - 1) input tile and that
//   plane's kH x kWp x 16 weights (k-major, zero padded to kWp = 4 ceil(kW/4)) are staged in LDS; per (ky, 4 kx) step a
//   wave reads one B fragment and, for each of its four 16-pixel tiles, one A fragment (overlapping lanes broadcast) + one
//   MFMA.  Zero-padded weights multiply finite tile values (the tile is zero-filled past the frame), so padding adds
//   exact zeros to the chain.
#include "dfe_internal.h"

#ifndef DFE_CM_STAMPS
#define DFE_CM_STAMPS 0
#endif
#ifndef DFE_CM_SCHED
#define DFE_CM_SCHED 1
#endif
#ifndef DFE_CM_ABL
#define DFE_CM_ABL 0   // tuning: 1 = the step loop without its LDS reads, 2 = no output stores, 3 = no staging of the next tile
#endif
namespace {

typedef float f4v __attribute__((ext_vector_type(4)));

template <bool TANH>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
                                                       int nIn, int nOut, int H, int W, int kH, int kW, float *__restrict__ out) {
    extern __shared__ float smem[];
    const int Ho = H - kH + 1, Wo = W - kW + 1;
    const int kWp = (kW + 3) & ~3;
    const int TW = 64 + kWp - 1 + 3, TH = kH + 3;      // tile: 4 output rows, 64 columns (+3: the last k-step's lanes read past kWp-1)
    float *tile = smem;                                // [TH][TW]
    float *wl = smem + TH * TW;                        // [kH][kWp][16]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 4, n0 = blockIdx.z * 16;
    const int m = lane & 15, kq = lane >> 4;
    const int n = n0 + m;
    f4v acc[4];
    {
        const float b = (bias && n < nOut) ? bias[n] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f4v{b, b, b, b};
    }
    for (int i = 0; i < nIn; ++i) {
        __syncthreads();
        for (int e = threadIdx.x; e < TH * TW; e += 256) {
            const int r = e / TW, c = e - r * TW;
            const int y = y0 + r, x = x0 + c;
            tile[e] = (y < H && x < W) ? in[((long long)i * H + y) * W + x] : 0.f;
        }
        for (int e = threadIdx.x; e < kH * kWp * 16; e += 256) {
            const int nn = e & 15, v = (e >> 4) % kWp, u = (e >> 4) / kWp;
            wl[e] = (v < kW && n0 + nn < nOut) ? w[(((long long)(n0 + nn) * nIn + i) * kH + u) * kW + v] : 0.f;
        }
        __syncthreads();
        for (int u = 0; u < kH; ++u) {
            const float *trow = tile + (u + wave) * TW + m + kq;
            const float *wrow = wl + (u * kWp + kq) * 16 + m;
            for (int v0 = 0; v0 < kWp; v0 += 4) {
                const float b = wrow[v0 * 16];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(trow[v0 + 16 * t], b, acc[t], 0, 0, 0);
            }
        }
    }
    const int y = y0 + wave;
    if (y < Ho && n < nOut) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int x = x0 + 16 * t + 4 * kq;        // D[m' = 4 kq + r][n = lane & 15]
            float *o = out + ((long long)n * Ho + y) * Wo + x;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (x + r < Wo) o[r] = TANH ? tanhf(acc[t][r]) : acc[t][r];
        }
    }
}

// ---- round 5: weights RESIDENT in LDS, K flattened, persistent blocks ------------------------------------------------------------
// The kernel above restages a plane's weights with every input tile, pads every kernel row to a multiple of 4 taps (17 -> 20: 15 % of
// the MFMAs multiply zeros), gives a block 16 output planes (version2's 32 planes: every tile staged twice) and runs at 61 TFLOP/s on
// version2's 17 x 17 x 3 -> 32 layer (0.246 ms per VGA frame -- no faster than the exact VALU kernel).  This one:
//   * K = (input plane, ky, kx) is ONE flat index, padded once at its end (867 -> 896 = 28 chunks of 8 steps): lane (m, kq) walks
//     k = 4 s + kq through the tile by a table of tap offsets in LDS (first version: a carried (v, u, offset) triple with compares
//     and selects per step -- their issue time exceeded the MFMAs': 0.455 ms for both VGA frames, no faster than the exact kernel);
//   * the whole weight matrix [K4][16][NT] (NT = 1 or 2 groups of 16 output planes, interleaved so that a lane's two B operands are
//     one 8-byte read) is staged ONCE per block: 111 KB for version2's layer;
//   * one block of 16 waves per CU walks tiles of 4 rows x 64 columns (both frames of a pair in one launch): wave w owns row w / 4
//     and the 16 pixels w % 4 of it for all output planes -- one A read feeds NT MFMAs; the next tile's pixels are loaded into
//     registers before the step loop and written to the other LDS buffer behind it (one barrier per tile).
// The chain per output is the same k-ordered fmaf chain (bias first, padded taps add exact zeros at the END of the chain): bit-exact
// against the oracle's fmaf variant, like the kernel above.
struct CmBatch {
    const float *in[2];
    float *out[2];
    int H[2], W[2];                 // input size of entry e (output: H - kH + 1, W - kW + 1)
    unsigned pitch[2], plane[2];    // floats between rows / planes of in[e] (a view is allowed)
    float *nrm[2];                  // or NULL: [Ho][Wo] sum over the output planes of out^2 (the matrix-core matcher's |a|^2, |b|^2: feat_matching_mfma.hip)
    int tx[2];                      // tiles per output row
    unsigned txm[2];                // ceil(2^32 / tx): by = (rel * txm) >> 32 on the scalar unit (exact: rel < 2^16)
    int t0[3];                      // first tile of entry e; t0[n] = number of tiles
    int n;
};

template <int NT, bool TANH>
__global__ __launch_bounds__(1024) void conv_mfma_res_kernel(CmBatch cb, const float *__restrict__ w, const float *__restrict__ bias, int nIn, int nOut, int kH,
                                                             int kW) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TR = 4, TC = 64;                                  // output rows / columns of a tile
    constexpr int CH = 8;                                           // steps per chunk: a chunk's LDS reads are all issued before its MFMAs
    const int K = nIn * kH * kW, steps = ((K + 3) / 4 + CH - 1) / CH * CH;
    const int TH = TR + kH - 1, TW = TC + kW - 1, TPL = TH * TW, TSZ = nIn * TPL;
    float *wl = smem;                                               // [4 steps][16][NT]
    int *ktab = reinterpret_cast<int *>(smem + (size_t)steps * 4 * 16 * NT);   // [4 steps]: BYTE offset of tap k inside a tile
    float *tile0 = reinterpret_cast<float *>(ktab + steps * 4);     // [2][nIn][TH][TW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    // ---- weights and the tap table, once: thread t takes the (plane, group) pair t % (16 NT) and every 1024 / (16 NT)-th k
    {
        constexpr int PP = 16 * NT;
        const int pn = tid % PP, mm = pn / NT, nt = pn - mm * NT, plane = nt * 16 + mm;
        for (int k = tid / PP; k < steps * 4; k += 1024 / PP)
            wl[k * PP + pn] = (k < K && plane < nOut) ? w[(long long)plane * K + k] : 0.f;
        // (the flat index k = (input plane, ky, kx) walks the tile in steps of 4 taps that straddle kernel rows and planes: the walk's
        //  per-lane carry chain -- compare / select per step -- cost more issue time than the MFMAs; a table read does not)
        for (int k = tid; k < steps * 4; k += 1024) {
            const int kk = k < K ? k : 0;                           // (padded taps: zero weights, any initialised element)
            const int i = kk / (kH * kW), r = kk - i * kH * kW, u = r / kW, v = r - u * kW;
            ktab[k] = 4 * (i * TPL + u * TW + v);
        }
    }
    // ---- this thread's share of a tile's staging: elements e = tid + 1024 j of [nIn][TH][TW]
    constexpr int NLD = 6;                                          // (launcher: nIn * TH * TW <= 6 * 1024)
    int er[NLD], ec[NLD];
    unsigned ep[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int e = tid + 1024 * j, i = e / TPL, rem = e - i * TPL;
        er[j] = rem / TW; ec[j] = rem - er[j] * TW;
        ep[j] = (unsigned)i;                                        // (times the entry's plane stride at the load)
    }
    // (everything here is wave-uniform: the multiply-high stands in for a division, which the compiler would run on the VECTOR unit even
    //  for uniform operands -- and next to f32 MFMAs every vector instruction costs its full issue time: the matrix pipe does not run
    //  beside it, profiles/r05_y_conv_mfma_stamps.txt)
    auto tile_geom = [&](int t, int &ent, int &x0, int &y0) {
        ent = (cb.n > 1 && t >= cb.t0[1]) ? 1 : 0;
        const int rel = t - cb.t0[ent], by = cb.tx[ent] == 1 ? rel : (int)__umulhi((unsigned)rel, cb.txm[ent]);   // (tx = 1: the reciprocal 2^32 has no 32-bit form)
        x0 = (rel - by * cb.tx[ent]) * TC; y0 = by * TR;
    };
    float stg[NLD];
    unsigned eo[2][NLD];                                            // element j's offset inside a tile of entry e (floats): interior tiles need no clamp
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int j = 0; j < NLD; ++j) eo[e][j] = ep[j] * cb.plane[e < cb.n ? e : 0] + (unsigned)er[j] * cb.pitch[e < cb.n ? e : 0] + (unsigned)ec[j];
    auto load_tile = [&](int t) {
        int ent, x0, y0;
        tile_geom(t, ent, x0, y0);
        const float *__restrict__ in = cb.in[ent];
        if (y0 + TH <= cb.H[ent] && x0 + TW <= cb.W[ent]) {          // (uniform) the tile and its halo lie inside the frame: scalar base + the thread's offsets
            const float *__restrict__ base = in + ((size_t)y0 * cb.pitch[ent] + x0);
#pragma unroll
            for (int j = 0; j < NLD; ++j)
                if (tid + 1024 * j < TSZ) stg[j] = base[ent ? eo[1][j] : eo[0][j]];
        } else {
#pragma unroll
            for (int j = 0; j < NLD; ++j)
                if (tid + 1024 * j < TSZ)
                    stg[j] = in[(size_t)ep[j] * cb.plane[ent] + (size_t)min(y0 + er[j], cb.H[ent] - 1) * cb.pitch[ent] + min(x0 + ec[j], cb.W[ent] - 1)];
        }
    };
    auto store_tile = [&](float *buf) {
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            if (tid + 1024 * j < TSZ) buf[tid + 1024 * j] = stg[j];
    };
    const int ntiles = cb.t0[cb.n];
    int t = blockIdx.x;
    if (t < ntiles) { load_tile(t); store_tile(tile0); }
    __syncthreads();
    const int row = wave >> 2, tsel = wave & 3;
    const int abase = row * TW + 16 * tsel + m;
    typedef float bv_t __attribute__((ext_vector_type(NT)));
    int cur = 0;
    for (; t < ntiles; t += gridDim.x) {
#if DFE_CM_STAMPS
        unsigned st_[5];
        st_[0] = (unsigned)__builtin_readcyclecounter();
#endif
        const int tn = t + gridDim.x;
        const int nch = steps / CH;
        if (tn < ntiles && DFE_CM_ABL != 3) load_tile(tn);          // (in flight behind the step loop)
        const char *tb = reinterpret_cast<const char *>(tile0 + cur * TSZ + abase);
        f4v acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int plane = nt * 16 + m;
            const float b = (bias && plane < nOut) ? bias[plane] : 0.f;
            acc[nt] = f4v{b, b, b, b};
        }
        const int *kp = ktab + kq;                                  // lane (m, kq) walks k = 4 s + kq
        const float *wp = wl + (kq * 16 + m) * NT;
        // Two register sets: a chunk's operands are requested one chunk AHEAD, behind the MFMAs of the chunk before it, so that a wave's
        // reads never wait in front of its own MFMAs.  (One set -- reads, wait, MFMAs -- left the matrix pipe idle whenever the four waves
        // of a SIMD, which leave every barrier in phase, all sat in their reads at once: 74 % MFMA-busy by the counters,
        // profiles/r05_y_pmc_version2_mfma.txt.)
        int off[CH] = {};
        float a0[CH], a1[CH];
        bv_t b0[CH], b1[CH];
        auto fetch = [&](float (&a)[CH], bv_t (&b)[CH]) {
#if DFE_CM_ABL == 1
#pragma unroll
            for (int j = 0; j < CH; ++j) { a[j] = __int_as_float(off[j]); for (int nt = 0; nt < NT; ++nt) b[j][nt] = __int_as_float(off[j] + nt); }
#else
#if DFE_CM_ABL == 6 || DFE_CM_ABL == 9     // A operands at immediate offsets: no address adds, no tap table
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = *reinterpret_cast<const float *>(tb + 64 * j);
#elif DFE_CM_ABL == 8                      // no A reads at all
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = __int_as_float(off[0]);
#else
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = *reinterpret_cast<const float *>(tb + off[j]);
#endif
#if DFE_CM_ABL == 7 || DFE_CM_ABL == 9     // no B reads
#pragma unroll
            for (int j = 0; j < CH; ++j) for (int nt = 0; nt < NT; ++nt) b[j][nt] = __int_as_float(off[0]);
#else
#pragma unroll
            for (int j = 0; j < CH; ++j) b[j] = *reinterpret_cast<const bv_t *>(wp + j * 4 * 16 * NT);
#endif
#endif
            wp += CH * 4 * 16 * NT;
            kp += 4 * CH;
        };
        auto taps = [&]() {
#if DFE_CM_ABL == 1
#pragma unroll
            for (int j = 0; j < CH; ++j) off[j] += j;
#elif DFE_CM_ABL == 6 || DFE_CM_ABL == 8 || DFE_CM_ABL == 9
#else
#pragma unroll
            for (int j = 0; j < CH; ++j) off[j] = kp[4 * j];
#endif
        };
        auto mm = [&](const float (&a)[CH], const bv_t (&b)[CH]) {
#if DFE_CM_ABL == 4
#pragma unroll
            for (int j = 0; j < CH; ++j) asm volatile("" ::"v"(a[j]), "v"(b[j]));
#else
#pragma unroll
            for (int j = 0; j < CH; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j][nt], acc[nt], 0, 0, 0);
#if DFE_CM_ABL == 5
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j][nt], acc[nt], 0, 0, 0);
#endif
                }
#endif
        };
        // (the scheduler is told to put one operand read and one address add behind every MFMA instead of all 24 reads in front of the
        //  sixteen MFMAs: a wave that runs ALONE on its SIMD -- the last one of a tile, the matrix pipe's arbitration is not fair -- then
        //  still keeps the pipe busy: DFE_CM_SCHED)
        auto interleave = [&]() {
#if DFE_CM_SCHED
#pragma unroll
            for (int q = 0; q < CH * NT; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // one VALU (an A address)
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read
            }
#endif
        };
        taps();
        fetch(a0, b0);                                               // chunk 0
        taps();                                                      // (one chunk past the end: reads initialised LDS, never used)
#if DFE_CM_STAMPS
        st_[1] = (unsigned)__builtin_readcyclecounter();
#endif
        int c = 0;
        for (; c + 2 < nch; c += 2) {                                // a0 / b0 hold chunk c
            fetch(a1, b1);
            taps();
            mm(a0, b0);
            interleave();
            fetch(a0, b0);
            taps();
            mm(a1, b1);
            interleave();
        }
        if (c + 1 < nch) {
            fetch(a1, b1);
            mm(a0, b0);
            mm(a1, b1);
        } else {
            mm(a0, b0);
        }
#if DFE_CM_STAMPS
        asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[NT - 1][3]));   // (the accumulators are complete)
        st_[2] = (unsigned)__builtin_readcyclecounter();
#endif
        {
            int ent, x0, y0;
            tile_geom(t, ent, x0, y0);
            const int Ho = cb.H[ent] - kH + 1, Wo = cb.W[ent] - kW + 1;
            const int y = y0 + row, x = x0 + 16 * tsel + 4 * kq;    // D[pixel 4 kq + r][plane = lane & 15]
            // (row, tsel, y and the tile's origin are wave-uniform: the output's base address is scalar arithmetic, a lane adds its plane and
            //  its four pixels as ONE 32-bit offset)
            const size_t obase = (size_t)y * Wo + (size_t)(x0 + 16 * tsel);
            const unsigned HoWo = (unsigned)Ho * (unsigned)Wo;
            if (cb.nrm[ent]) {
                // the pixels' squared norms over the output planes, for the matcher that follows: this lane's planes first, then the 16 lanes
                // of its DPP row (partners at distance 8, 4, 2, 1); one lane per four pixels stores them
                f4v sq = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (nt * 16 + m < nOut) {
                        f4v r = acc[nt];
                        if constexpr (TANH) r = f4v{tanhf(r[0]), tanhf(r[1]), tanhf(r[2]), tanhf(r[3])};
#pragma unroll
                        for (int q = 0; q < 4; ++q) sq[q] = fmaf(r[q], r[q], sq[q]);
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q) sq[q] = row16_sum_f32_ordered(sq[q]);
                if (m == 0 && y < Ho) {
                    float *o = cb.nrm[ent] + obase + 4u * (unsigned)kq;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (x + q < Wo) o[q] = sq[q];
                }
            }
            if (y < Ho && (DFE_CM_ABL != 2 || acc[0][0] == 1234.5f)) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int plane = nt * 16 + m;
                    if (plane >= nOut) continue;
                    float *o = cb.out[ent] + obase + ((unsigned)plane * HoWo + 4u * (unsigned)kq);
                    f4v r = acc[nt];
                    if constexpr (TANH) r = f4v{tanhf(r[0]), tanhf(r[1]), tanhf(r[2]), tanhf(r[3])};
                    if (x + 3 < Wo && !(((uintptr_t)o) & 15)) *reinterpret_cast<f4v *>(o) = r;
                    else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (x + q < Wo) o[q] = r[q];
                    }
                }
            }
        }
        if (tn < ntiles && DFE_CM_ABL != 3) store_tile(tile0 + (cur ^ 1) * TSZ);
        // LDS-only barrier: __syncthreads() also drains vmcnt, i.e. every wave would wait here until the tile's output stores are
        // acknowledged by the memory -- with nothing on the matrix pipe meanwhile
#if DFE_CM_STAMPS
        st_[3] = (unsigned)__builtin_readcyclecounter();
#endif
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        cur ^= 1;
#if DFE_CM_STAMPS
        st_[4] = (unsigned)__builtin_readcyclecounter();
        {
            const int it = (t - (int)blockIdx.x) / (int)gridDim.x;
            if (blockIdx.x == 3 && lane == 0 && it < 8) {
                unsigned *dbg = reinterpret_cast<unsigned *>(cb.out[0]);
                for (int q = 0; q < 5; ++q) dbg[(it * 16 + wave) * 5 + q] = st_[q];
            }
        }
#endif
    }
}

}  // namespace

static size_t conv_mfma_res_lds(int nIn, int nOut, int kH, int kW) {
    const int NT = nOut > 16 ? 2 : 1, steps = ((nIn * kH * kW + 3) / 4 + 7) / 8 * 8;
    return ((size_t)steps * 4 * 16 * NT + (size_t)steps * 4 + (size_t)2 * nIn * (4 + kH - 1) * (64 + kW - 1)) * sizeof(float);
}
// n <= 2 inputs (the two frames of a pair) through ONE layer in one launch; *handled = false: not this kernel's shape
int dfe_conv_mfma_res_batch(dfe_ctx *ctx, int n, const float *const *in, const int *H, const int *W, const int *in_pitch, const long long *in_plane,
                            const dfe_filter_layer &L, float *const *out, bool *handled, float *const *nrm) {
    *handled = false;
    if (n < 1 || n > 2 || L.conn || L.nOut > 32) return DFE_OK;
    const size_t lds = conv_mfma_res_lds(L.nIn, L.nOut, L.kH, L.kW);
    if (lds > 160 * 1024 || (size_t)L.nIn * (4 + L.kH - 1) * (64 + L.kW - 1) > 6 * 1024) return DFE_OK;
    CmBatch cb{};
    cb.n = n;
    int nt = 0;
    for (int e = 0; e < n; ++e) {
        if (H[e] < L.kH || W[e] < L.kW) return DFE_OK;
        const long long pit = in_pitch ? in_pitch[e] : W[e], pla = in_plane ? in_plane[e] : (long long)H[e] * W[e];
        if (pit >= (1ll << 31) || pla >= (1ll << 31)) return DFE_OK;
        cb.in[e] = in[e]; cb.out[e] = out[e]; cb.H[e] = H[e]; cb.W[e] = W[e]; cb.pitch[e] = (unsigned)pit; cb.plane[e] = (unsigned)pla;
        cb.nrm[e] = nrm ? nrm[e] : nullptr;
        cb.tx[e] = dfe_cdiv(W[e] - L.kW + 1, 64);
        cb.txm[e] = cb.tx[e] == 1 ? 0u : (unsigned)(((1ull << 32) + cb.tx[e] - 1) / cb.tx[e]);
        if ((long long)cb.tx[e] * dfe_cdiv(H[e] - L.kH + 1, 4) >= 65536) return DFE_OK;   // (the kernel's multiply-high division is exact below that)
        cb.t0[e] = nt;
        nt += cb.tx[e] * dfe_cdiv(H[e] - L.kH + 1, 4);
    }
    cb.t0[n] = nt;
    const int grid = nt < ctx->ncu ? nt : ctx->ncu;
    auto kern = L.nOut > 16 ? (L.tanh_after ? conv_mfma_res_kernel<2, true> : conv_mfma_res_kernel<2, false>)
                            : (L.tanh_after ? conv_mfma_res_kernel<1, true> : conv_mfma_res_kernel<1, false>);
    DFE_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        DfeProfScope prof(ctx);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, ctx->stream, cb, L.weight, L.bias, L.nIn, L.nOut, L.kH, L.kW);
    }
    DFE_LAUNCH_CHECK(ctx);
    ctx->last_kernel = "conv_mfma_res_kernel";
    *handled = true;
    return DFE_OK;
}

size_t conv_mfma_lds_bytes(int kH, int kW) {
    const int kWp = (kW + 3) & ~3;
    return ((size_t)(kH + 3) * (64 + kWp - 1 + 3) + (size_t)kH * kWp * 16) * sizeof(float);
}

int dfe_conv_mfma_launch(dfe_ctx *ctx, const float *in, const float *weight, const float *bias, int nIn, int nOut, int H, int W, int kH, int kW,
                         int tanh_after, float *out, bool *handled) {
    *handled = false;
    {   // the resident-weights kernel where the layer's weight matrix fits LDS next to two input tiles
        dfe_filter_layer L{};
        L.nIn = nIn; L.nOut = nOut; L.kH = kH; L.kW = kW; L.weight = weight; L.bias = bias; L.tanh_after = tanh_after;
        int rc = dfe_conv_mfma_res_batch(ctx, 1, &in, &H, &W, nullptr, nullptr, L, &out, handled, nullptr);
        if (rc || *handled) return rc;
    }
    const size_t lds = conv_mfma_lds_bytes(kH, kW);
    if (lds > 96 * 1024) return DFE_OK;
    const int Ho = H - kH + 1, Wo = W - kW + 1;
    dim3 grid(dfe_cdiv(Wo, 64), dfe_cdiv(Ho, 4), dfe_cdiv(nOut, 16));
    if (tanh_after) {
        DFE_HIP(ctx, hipFuncSetAttribute((const void *)conv_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(conv_mfma_kernel<true>, grid, dim3(256), lds, ctx->stream, in, weight, bias, nIn, nOut, H, W, kH, kW, out);
    } else {
        DFE_HIP(ctx, hipFuncSetAttribute((const void *)conv_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(conv_mfma_kernel<false>, grid, dim3(256), lds, ctx->stream, in, weight, bias, nIn, nOut, H, W, kH, kW, out);
    }
    DFE_LAUNCH_CHECK(ctx);
    ctx->last_kernel = "conv_mfma_kernel";
    *handled = true;
    return DFE_OK;
}

extern "C" {

int dfe_spatial_convolution_mfma_f32(dfe_ctx *ctx, const float *in, const float *weight, const float *bias, int nIn, int nOut, int H, int W,
                                     int kH, int kW, int tanh_after, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, in && weight && out, DFE_E_ARG, "dfe_spatial_convolution_mfma_f32: NULL tensor");
    DFE_REQUIRE(ctx, nIn > 0 && nOut > 0 && kH > 0 && kW > 0 && H >= kH && W >= kW, DFE_E_SHAPE,
                "dfe_spatial_convolution_mfma_f32: %d->%d planes, %dx%d kernel on %dx%d", nIn, nOut, kH, kW, H, W);
    bool handled = false;
    int rc = dfe_conv_mfma_launch(ctx, in, weight, bias, nIn, nOut, H, W, kH, kW, tanh_after, out, &handled);
    if (rc) return rc;
    DFE_REQUIRE(ctx, handled, DFE_E_UNSUPPORTED, "dfe_spatial_convolution_mfma_f32: a %dx%d kernel needs %zu bytes of LDS per block", kH, kW,
                conv_mfma_lds_bytes(kH, kW));
    return DFE_OK;
}

}  // extern "C"
