// dfe_internal.h -- shared by the translation units of libdfe.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/dfe.h"

struct dfe_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int cv_mode = 0;                  // dfe_set_cost_volume_kernel
    int cv_chunk0 = 0;                // tiled kernel covers chunks >= this (set by the row-image launcher for its tail)
    int cv_tyq = 0;                   // 0 = pick the tile height per shape; else dfe_set_cost_volume_tile's code (tuning / tests)
    int ncu = 256;                    // compute units of the device
    const char *last_kernel = "";
    void *scratch = nullptr;          // grow-only device arena (never shrinks; freed with the ctx)
    size_t scratch_bytes = 0;
    size_t scratch_limit = (size_t)16 << 30;   // cost-volume bands are sized to fit (dfe_set_scratch_limit)
    int *dflag = nullptr;             // one device int for error flags raised by kernels
    char err[512] = {0};
    // optional per-launch timing of the cost-volume kernel (dfe_profile_enable)
    bool profile = false;
    int prof_depth = 0;               // only the outermost DfeProfScope records (a launcher may nest another)
    std::vector<hipEvent_t> prof_events;   // start/stop pairs, resolved by dfe_profile_read
    // launch-bound one-call pipelines (the multiscale matcher: 4-6 launches of 9-33 us) can be replayed as a hipGraph when
    // they are called again with the same arguments (same buffers, same shapes): the second such call captures, later ones
    // replay.  OFF unless DFE_GRAPHS=1 is in the environment when the ctx is created: on ROCm 7.2 / MI355X the replay
    // measured SLOWER than the direct launches (VGA 3-level pyramid 0.0857 against 0.0803 ms per pair, 1080p 0.543 against
    // 0.537 ms: the graph launch costs more than four back-to-back kernel launches on one stream).  Profiling and the
    // legacy null stream (not capturable) also turn it off.
    // stage timers with the reference's names (depth_estimation_opticalflow.lua:144-148: load / filter / match / extract): event
    // pairs per stage on the ctx stream, resolved by dfe_stage_timers_read
    bool stage_timers = false;
    int stage_depth = 0;                     // only the outermost DfeStageScope records
    struct StageEvent { int stage; hipEvent_t a, b; };
    std::vector<StageEvent> stage_events;
    struct GraphSlot {
        std::vector<unsigned char> key;
        int hits = 0;
        hipGraphExec_t exec = nullptr;
    };
    GraphSlot ms_graph;
    bool graphs = false;
};
// key = the bytes of a POD describing the call; returns 0 = launch directly, 1 = capture this call, 2 = replay slot.exec
int dfe_graph_lookup(dfe_ctx *ctx, dfe_ctx::GraphSlot &slot, const void *key, size_t bytes);
// ends a capture begun after dfe_graph_lookup returned 1 (rc = the launcher's result), instantiates and launches the graph
int dfe_graph_finish(dfe_ctx *ctx, dfe_ctx::GraphSlot &slot, int rc);

// brackets the cost-volume kernel launch with events on the ctx stream when profiling is on
struct DfeProfScope {
    dfe_ctx *ctx;
    explicit DfeProfScope(dfe_ctx *c) : ctx(c) {
        if (ctx->profile && ctx->prof_depth++ == 0) {
            hipEvent_t e;
            if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, ctx->stream); ctx->prof_events.push_back(e); }
        }
    }
    ~DfeProfScope() {
        if (ctx->profile && --ctx->prof_depth == 0 && (ctx->prof_events.size() & 1)) {
            hipEvent_t e;
            if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, ctx->stream); ctx->prof_events.push_back(e); }
        }
    }
};

// brackets the launches of one pipeline stage (DFE_STAGE_*) with events when the stage timers are on
struct DfeStageScope {
    dfe_ctx *ctx;
    bool rec = false;
    dfe_ctx::StageEvent ev{};
    DfeStageScope(dfe_ctx *c, int stage) : ctx(c) {
        if (ctx->stage_timers && ctx->stage_depth++ == 0) {
            ev.stage = stage;
            rec = hipEventCreate(&ev.a) == hipSuccess && hipEventCreate(&ev.b) == hipSuccess;
            if (rec) (void)hipEventRecord(ev.a, ctx->stream);
        }
    }
    ~DfeStageScope() {
        if (ctx->stage_timers) --ctx->stage_depth;
        if (rec) { (void)hipEventRecord(ev.b, ctx->stream); ctx->stage_events.push_back(ev); }
    }
    DfeStageScope(const DfeStageScope &) = delete;
    DfeStageScope &operator=(const DfeStageScope &) = delete;
};

int dfe_fail(dfe_ctx *ctx, int code, const char *fmt, ...);
// cost volume of raw frames into `out` (ssd_cost_volume.hip); H = rows visible to this call, plane = channel stride
int cv_frames_dispatch(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, long long plane, int kh,
                       int kw, int hWin, int wWin, float *out);
// the volumes of n independent pairs (pyramid scales) in one launch where a common block shape exists (*handled)
// prob (may be NULL): per pair, non-null = leave soft-min probabilities there instead of the costs in out[i] -- if the
// launcher finds that worthwhile for the shape (*prob_used)
int cv_frames_dispatch_multi(dfe_ctx *ctx, int n, const float *const *I0, const float *const *I1, int C, const int *H, const int *W, int k,
                             int hWin, int wWin, float *const *out, float *const *prob, bool *handled, bool *prob_used, float f16_scale = 0.f, int nq_hint = 0);
// The finest scale of the multiscale matcher WITHOUT its volume (multiscale.hip -> ssd_cost_volume.hip): the tiled kernel's task rows
// (8 pixels x the 64 cells of an 8 x 8 window, lane <-> cell) go through soft-min, cascade add, arg-max and decode in registers
struct CvFineArgs {
    const float *pcasc;        // cascaded windows of the next coarser scale [H/2][W/2][64], or NULL (one ratio only)
    const float2 *pbest;       // its running best (value, 0-based class as int bits)
    long long *idx;            // out [H][W]: 1-based class ids, or NULL
    float *fy, *fx;            // out [H][W]: decoded displacement planes, or NULL
    int middle;                // centre class (yx2xMulti(0, 0)), 1-based
    float f16_scale, f16_inv;  // != 0: the costs are rounded to half precision (cost * scale) first, as a stored fp16 volume would be
    int dec[5 * 64];           // class id - 1 -> (oy << 16) | (ox & 0xffff)
    // a scale > 1 (the same epilogue up to the cascade add, then what cascade_px_kernel<false> leaves for the next finer scale):
    float *casc;               // != NULL: out [H][W][64] cascaded windows of THIS scale; idx / fy / fx unused
    float2 *best;              //          out [H][W] running best (value, 0-based class as int bits)
    int cls_base;              //          0-based class id of this scale's first ring cell
};
int cv_frames_finest_fused(dfe_ctx *ctx, const float *I0p, const float *I1p, int C, int Hp, int Wp, int k, int maxh, int maxw, const CvFineArgs &fine,
                           bool *handled);
bool cv_finest_plan_ok(dfe_ctx *ctx, int Hp, int Wp, int maxh, int maxw);   // cv_frames_finest_fused (with a parent scale) would take this frame
int dfe_scratch(dfe_ctx *ctx, size_t bytes, void **out);   // arena of at least `bytes`
// one layer of a filter stack (filters.hip): in [nIn][H][W] -> out [nOut][H-kH+1][W-kW+1], nn.Tanh fused behind it when
// L.tanh_after (the same tanhf as dfe_tanh_f32: bit-identical to the two separate calls)
int dfe_filter_layer_forward(dfe_ctx *ctx, const float *in, const dfe_filter_layer &L, int H, int W, float *out);
// the same layer position of n independent inputs (both frames of every pyramid scale) in ONE launch where a batched kernel exists
int dfe_filter_layer_forward_batch(dfe_ctx *ctx, int n, const float *const *in, const dfe_filter_layer *const *L, const int *H, const int *W,
                                   float *const *out);
// nn.SpatialMatching on feature maps, fast kernels or the reference-order one (ssd_cost_volume.hip)
int dfe_spatial_matching_dispatch(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh, int maxw, float *out);

#define DFE_HIP(ctx, expr)                                                              \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return dfe_fail((ctx), DFE_E_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                        \
    } while (0)

#define DFE_REQUIRE(ctx, cond, code, ...)                    \
    do {                                                     \
        if (!(cond)) return dfe_fail((ctx), (code), __VA_ARGS__); \
    } while (0)

#define DFE_LAUNCH_CHECK(ctx) DFE_HIP(ctx, hipGetLastError())

// Every extern "C" entry point runs on ITS ctx's device whatever the caller's current device is (one ctx per GPU, several
// ctxs per host thread are legal: include/dfe.h), and leaves the caller's current device as it found it.
struct DfeDeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DfeDeviceGuard(const dfe_ctx *c) {
        if (c && hipGetDevice(&prev) == hipSuccess && prev != c->device) switched = hipSetDevice(c->device) == hipSuccess;
    }
    ~DfeDeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DfeDeviceGuard(const DfeDeviceGuard &) = delete;
    DfeDeviceGuard &operator=(const DfeDeviceGuard &) = delete;
};
// first statement of an entry point: NULL check + device guard for the rest of the call
#define DFE_ENTER(ctx)                                          \
    DFE_REQUIRE((ctx), (ctx), DFE_E_ARG, "ctx is NULL");        \
    DfeDeviceGuard dfe_device_guard_(ctx)


static inline int dfe_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }


// ---- shared by the fused cost-volume epilogue (ssd_cost_volume.hip) and the tail kernels (postops.hip) ----
// sorting networks of extract_output.cpp:17-61: a comparator swaps value AND index iff v[b] > v[a]
__device__ __forceinline__ void dfe_sortswap(float *v, float *ix, int a, int b) {
    if (v[b] > v[a]) {
        float t = v[b]; v[b] = v[a]; v[a] = t;
        t = ix[b]; ix[b] = ix[a]; ix[a] = t;
    }
}
__device__ __forceinline__ void dfe_sort4(float *v, float *ix) {   // :27-33
    dfe_sortswap(v, ix, 0, 2); dfe_sortswap(v, ix, 1, 3); dfe_sortswap(v, ix, 0, 1); dfe_sortswap(v, ix, 2, 3); dfe_sortswap(v, ix, 1, 2);
}
__device__ __forceinline__ void dfe_sort8(float *v, float *ix) {   // :35-61
    dfe_sortswap(v, ix, 0, 1); dfe_sortswap(v, ix, 2, 3); dfe_sortswap(v, ix, 4, 5); dfe_sortswap(v, ix, 6, 7);
    dfe_sortswap(v, ix, 0, 2); dfe_sortswap(v, ix, 1, 3); dfe_sortswap(v, ix, 4, 6); dfe_sortswap(v, ix, 5, 7);
    dfe_sortswap(v, ix, 1, 2); dfe_sortswap(v, ix, 5, 6); dfe_sortswap(v, ix, 0, 4); dfe_sortswap(v, ix, 3, 7);
    dfe_sortswap(v, ix, 1, 5); dfe_sortswap(v, ix, 2, 6);
    dfe_sortswap(v, ix, 1, 4); dfe_sortswap(v, ix, 3, 6);
    dfe_sortswap(v, ix, 2, 4); dfe_sortswap(v, ix, 3, 5);
    dfe_sortswap(v, ix, 3, 4);
}

// what the cost-volume kernel's fused epilogue leaves behind for flow_finalize_kernel
#define DFE_LEAD 16   // leading cells of every pixel kept for extractOutput
struct CvFuseArgs {
    float2 *part;          // [nchunks][Ptot]: per (chunk, pixel) the chunk's minimum cost and the 0-based index (int bits) of
                           // the first cell that attains it
    float *centre;         // [Ptot]: cost of the centre cell
    float *lead;           // [Ptot][DFE_LEAD]: the pixel's first cells
    long long Ptot;
    int cmid, lmid;        // chunk / lane of the centre cell
    int row_off;           // output-row offset of this launch inside the pair
    float *rec;            // the role-split row-image kernels leave their per-pixel results here instead of in the planes above:
                           // [column group = tile column][output row of the pair][DFE_REC floats] -- one 128-B line per TILE ROW:
                           // 8 x (minimum, first index as int bits) | 8 x centre cost | 8 x 0 -- written whole by ONE store of one wave,
                           // a block sweeping down its column writes consecutive lines.  (The planes took four partial-line stores per
                           // PIXEL and row step -- 8 B + 8 B + 64 B + 4 B, lines shared with neighbouring blocks on other XCDs -- and made
                           // the fused 1080p kernel take 1.78 .. 2.21 ms depending on the process; a 128-B record per pixel, 251 MB at
                           // 1080p, cost 0.8 ms: DESIGN section 5.)  The pixel's first DFE_LEAD cells are read back from the volume.
    int rec_rows;          // output rows of the pair (the record's row pitch)
};
#define DFE_REC 32         // floats per tile-row record
#define DFE_REC_CENTRE 16  // (entries 0..15: (minimum, index) of the 8 pixels; 16..23: their centre costs)
// frame mode of dfe_flow_finalize (one band only): finalize also zeroes the frame border and makes depth / confidence
struct DfePairDepth { int H, W; float cx, cy; float *depth, *conf; };
// (rec != nullptr: part / centre / lead are ignored -- minimum / index / centre come from the tile-row records [col group][rec_rows][DFE_REC],
//  the lead cells from the volume itself)
int dfe_flow_finalize(dfe_ctx *ctx, const float2 *part, const float *centre, const float *lead, int nchunks, long long Ptot,
                      const float *vol, double threshold, int rows, int Wo, int hWin, int wWin, int row_off, int64_t *idx, float *best,
                      float *fy, float *fx, float *scores, int64_t *imaxs, int pitch, int pad_t, int pad_l, int scores_padded,
                      const struct DfePairDepth *pd = nullptr, const float *rec = nullptr, int rec_rows = 0);
int dfe_pair_border_depth(dfe_ctx *ctx, float *flow, float *scores, int H, int W, int pad_t, int pad_l, int Ho, int Wo, float cx,
                          float cy, float *depth, float *conf);
int dfe_feat_matching_fast(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh, int maxw, float *out,
                           bool *handled);
int dfe_feat_matching_win64_batch(dfe_ctx *ctx, int n, const float *const *in1, const float *const *in2, int K, const int *H1, const int *W1, int maxh,
                                  int maxw, float *const *out, float f16_scale, bool *handled);
int cv_frames_dispatch_fused(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, long long plane, int k, int hWin,
                             int wWin, float *out, const CvFuseArgs &fa, bool *handled, int *nparts, bool *recs = nullptr);

// multiscale class-id geometry and decode, shared by postops.hip (x2yxMulti) and multiscale.hip (fused cascade -> flow)
struct MultiGeom {
    int maxh, maxw, nratios;
    int ratios[DFE_MAX_RATIOS];
    int d[DFE_MAX_RATIOS];   // ring width per scale (index >= 1)
};

// (I = int on the device's per-block decode tables: a 64-bit divide costs ~100 instructions there)
template <class I> __host__ __device__ inline int multi_decode_t(const MultiGeom &g, I id, I *oy, I *ox) {
    // replaces: x2yxMultiNumber opticalflow_model_multiscale.lua:83-132
    const int maxh = g.maxh, maxw = g.maxw;
    const int chh = (maxh + 1) / 2, chw = (maxw + 1) / 2;
    I x = id;
    if (x < 1) return -1;
    if (x <= (I)maxh * maxw) {
        *oy = (x - 1) / maxw + 1 - chh;
        *ox = (x - 1) % maxw + 1 - chw;
        return 0;
    }
    x -= (I)maxh * maxw;
    for (int i = 1; i < g.nratios; ++i) {
        const int d = g.d[i];
        const I len = (I)2 * d * maxw + (I)2 * (maxh - 2 * d) * d;
        I ty, tx;
        if (x <= len) {
            if (x <= (I)d * maxw) {
                ty = (x - 1) / maxw + 1; tx = (x - 1) % maxw + 1;
            } else {
                x -= (I)d * maxw;
                if (x <= (I)(maxh - 2 * d) * d) {
                    ty = (x - 1) / d + 1 + d; tx = (x - 1) % d + 1;
                } else {
                    x -= (I)(maxh - 2 * d) * d;
                    if (x <= (I)(maxh - 2 * d) * d) {
                        ty = (x - 1) / d + 1 + d; tx = (x - 1) % d + 1 + maxw - d;
                    } else {
                        x -= (I)(maxh - 2 * d) * d;
                        if (x > (I)d * maxw) return -1;
                        ty = (x - 1) / maxw + 1 + maxh - d; tx = (x - 1) % maxw + 1;
                    }
                }
            }
            *oy = (ty - chh) * g.ratios[i];
            *ox = (tx - chw) * g.ratios[i];
            return 0;
        }
        x -= len;
    }
    return -1;
}

__host__ __device__ inline int multi_decode(const MultiGeom &g, long long id, long long *oy, long long *ox) {
    return multi_decode_t<long long>(g, id, oy, ox);
}

#ifdef __HIPCC__
// All-lanes MINIMUM of EIGHT ints per lane (one per column) in ~40 VALU ops instead of 8 x 6 steps: a halving
// butterfly -- after exchanging with lane^1, lane^2, lane^4 each lane is left with the single column (lane & 7), then
// the row rotate by 8 and gfx950's v_permlane16_swap / v_permlane32_swap finish it.  Everything stays on the VALU (DPP
// quad permutes / row rotates fold into v_min_i32_dpp; no LDS crossbar).  On return every lane holds the wave minimum
// of column (lane & 7).
template <int TX> __device__ __forceinline__ int wave_min8(const int (&k)[TX], int lane) {
    static_assert(TX == 8, "butterfly is written for 8 columns");
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    int a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int mine = b0 ? k[2 * i + 1] : k[2 * i], other = b0 ? k[2 * i] : k[2 * i + 1];
        a[i] = min(mine, __builtin_amdgcn_update_dpp(0, other, 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int mine = b1 ? a[2 * i + 1] : a[2 * i], other = b1 ? a[2 * i] : a[2 * i + 1];
        b[i] = min(mine, __builtin_amdgcn_update_dpp(0, other, 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
    }
    int c;
    {
        // row_ror:4 -- quad q takes from quad q+1 (mod 4): even quads (bit2 = 0) read an odd quad and vice versa, and
        // the following ror:8 completes the row whichever neighbour was used.
        const int mine = b2 ? b[1] : b[0], other = b2 ? b[0] : b[1];
        c = min(mine, __builtin_amdgcn_update_dpp(0, other, 0x124, 0xf, 0xf, false));
    }
    c = min(c, __builtin_amdgcn_update_dpp(0, c, 0x128, 0xf, 0xf, false));                  // row_ror:8
    {   // gfx950 lane-swap instructions keep the cross-row steps on the VALU (no LDS-pipe swizzle/bpermute)
        const auto r = __builtin_amdgcn_permlane16_swap(c, c, false, false);                // rows {0,1} and {2,3} pair up
        c = min((int)r[0], (int)r[1]);
        const auto q = __builtin_amdgcn_permlane32_swap(c, c, false, false);                // halves pair up
        c = min((int)q[0], (int)q[1]);
    }
    return c;
}

// exp(x) for x <= 0 -- the soft-min's arguments, -c - max(-c): v_exp_f32 on x * log2(e), two instructions.  The product's rounding
// moves the result by |x| * 2^-24 relative at most, i.e. by less than 4e-8 ABSOLUTE for every x <= 0 (|x| e^x <= 1/e), against the
// 1e-6 the soft-min is held to (the reference's own nn.SoftMax of that era used a polynomial exp: its numerics are
// unpinned anyway).  Round 2 had the library expf without its overflow branch (Cody-Waite reduction + ldexp: 7 instructions, 13 in
// the library form) -- the 64 calls per pixel were 60 % of the finest cascade kernel's arithmetic.  EVERY soft-min on the device goes
// through this function, so the staged and the one-call paths stay bit-identical to each other.
__device__ __forceinline__ float dfe_exp_nonpos(float x) {
    return __builtin_amdgcn_exp2f(x * 0x1.715476p+0f);
}

// wave reductions of the soft-min (multiscale.hip and the volume kernel's soft-min epilogue): everything on the VALU
__device__ __forceinline__ float wave_max_f32(float v) {
#define DFE_STEP(ctrl) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false)))
    DFE_STEP(0xB1); DFE_STEP(0x4E); DFE_STEP(0x124); DFE_STEP(0x128);   // quad_perm [1,0,3,2], [2,3,0,1], row_ror:4, row_ror:8
#undef DFE_STEP
    const int b = __float_as_int(v);
    const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    v = fmaxf(__int_as_float((int)r[0]), __int_as_float((int)r[1]));
    const int c = __float_as_int(v);
    const auto q = __builtin_amdgcn_permlane32_swap(c, c, false, false);
    return fmaxf(__int_as_float((int)q[0]), __int_as_float((int)q[1]));
}
// wave sum in the association order of `for (off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off)` -- partners at
// distance 32, 16, 8, 4, 2, 1 -- on the VALU only (lane swaps + DPP), bit-identical to the shuffle version: after the
// distance-8 step lanes L and L^8 hold equal values, so row_ror:4 (partner (L+4) mod 16) reads the same number as L^4
__device__ __forceinline__ float wave_sum_f32_ordered(float v) {
    int b = __float_as_int(v);
    const auto q = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    v = __int_as_float((int)q[0]) + __int_as_float((int)q[1]);
    b = __float_as_int(v);
    const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    v = __int_as_float((int)r[0]) + __int_as_float((int)r[1]);
#define DFE_STEP(ctrl) v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false))
    DFE_STEP(0x128); DFE_STEP(0x124); DFE_STEP(0x4E); DFE_STEP(0xB1);   // row_ror:8, row_ror:4, quad_perm [2,3,0,1], [1,0,3,2]
#undef DFE_STEP
    return v;
}

#endif  // __HIPCC__

