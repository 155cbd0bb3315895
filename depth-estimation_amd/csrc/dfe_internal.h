// dfe_internal.h -- shared by the translation units of libdfe.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include <hip/hip_ext.h>
#include "../../include/dfe.h"

// Behaviour switches of the launchers (dfe_set_option / dfe_get_option, include/dfe.h).  -1 = automatic: the launcher's own choice
// per shape.  The environment is read ONCE, in dfe_ctx_create (tuning scripts), never inside a launcher.
enum DfeOpt {
    DFE_OPT_CASCADE_PX = 0,   // lane <-> pixel cascade kernels (0: keep the lane <-> cell kernels)
    DFE_OPT_FINE_FUSE,        // finest pyramid scale inside its volume / matcher kernel (0 / 1 forces)
    DFE_OPT_MID_FUSE,         // the second scale the same way
    DFE_OPT_FINE_NQ,          // tile-height code of the fused finest kernel
    DFE_OPT_MID_NQ,           // ... of the fused second scale
    DFE_OPT_PREP_TILES,       // every scale from one read of the frames (0: one launch per scale set)
    DFE_OPT_XPOSE,            // 1-KB transposed stores of one-chunk windows (0: 256-B pieces)
    DFE_OPT_XPOSE_NT,         // their non-temporal hint (0 / 1 forces)
    DFE_OPT_SOFT_EPILOGUE,    // soft-min inside the pyramid's volume kernel (0 / 1 forces)
    DFE_OPT_CONV_BATCH,       // batched LDS-tiled convolution (0: one direct launch per layer and input)
    DFE_OPT_CONV_NT10,        // 10 output planes per thread where nOut % 10 == 0 (0: two groups of 5)
    DFE_OPT_FM64,             // one-chunk feature matcher (0: chunk kernel per scale)
    DFE_OPT_FM_ROWS,          // feature matcher: the row kernels (0 / 1 forces)
    DFE_OPT_SWEEP_OVH,        // cost model of the persistent column sweep: per-piece overhead in rows
    DFE_OPT_SWEEP_BLOCKS,     // ... number of blocks
    DFE_OPT_DEBUG_ARENA,      // print where the scratch arena lands
    DFE_OPT_FM_FLAT,          // feature matcher: the flat-tile kernel for 16- / 17-wide windows (0: the round-3 row / chunk kernels)
    DFE_OPT_FM_SPLIT,         // ... a tile's window rows dealt to two co-resident half blocks (0: one block per tile)
    DFE_OPT_CONV_NARROW,      // batched convolution: 64 x 16 output tiles (0: 128 x 8; automatic: for kernels of 9 x 9 and larger)
    DFE_OPT_CONV_MFMA,        // one-call models: filter layers as implicit GEMMs on the matrix cores (fused multiply-adds; default 0 = exact kernels)
    DFE_OPT_FM_MFMA,          // feature matcher as a banded GEMM on the matrix cores, |a|^2 + |b|^2 - 2 a.b (default 0 = exact k-ordered sums)
    DFE_OPT_ARENA_CONTIG,     // scratch arena from physically contiguous memory (hipDeviceMallocContiguous; 0: plain hipMalloc)
    DFE_NOPT
};
struct DfeOptName { const char *key; const char *env; bool env_presence_means_zero; };
extern const DfeOptName dfe_opt_names[DFE_NOPT];

constexpr int DFE_NSLOT = 3;   // device slots of the pipelined ingest (ingest.hip)
struct dfe_ctx {
    int opt[DFE_NOPT];
    dfe_ctx() { for (int i = 0; i < DFE_NOPT; ++i) opt[i] = -1; }
    // the launcher's own choice `autov` unless the option was set (>= 0)
    bool opt_bool(int o, bool autov) const { return opt[o] < 0 ? autov : opt[o] != 0; }
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int cv_mode = 0;                  // dfe_set_cost_volume_kernel
    int cv_chunk0 = 0;                // tiled kernel covers chunks >= this (set by the row-image launcher for its tail)
    int cv_tyq = 0;                   // 0 = pick the tile height per shape; else dfe_set_cost_volume_tile's code (tuning / tests)
    int ncu = 256;                    // compute units of the device
    const char *last_kernel = "";
    void *scratch = nullptr;          // grow-only device arena (never shrinks; freed with the ctx), physically contiguous where the driver grants it
    size_t scratch_bytes = 0;
    void *scratch_plain = nullptr;    // the arena of the paths that run learned filter stacks: a plain hipMalloc (dfe_scratch's `plain`)
    size_t scratch_plain_bytes = 0;
    size_t scratch_limit = (size_t)16 << 30;   // cost-volume bands are sized to fit (dfe_set_scratch_limit)
    void *ingest = nullptr;           // grow-only fp32 copy of a uint8 frame pair (ingest.hip), freed with the ctx
    size_t ingest_bytes = 0;
    // pipelined ingest (ingest.hip): a copy stream of the ctx's own and DFE_NSLOT device slots for frame pairs -- the upload (+ conversion)
    // of pair i+1 runs beside the step of pair i; copied[s] / consumed[s] order the two streams per slot
    hipStream_t copy_stream = nullptr;
    void *slot[DFE_NSLOT] = {};        // two uint8 frames each
    size_t slot_bytes = 0;            // bytes of ONE frame of a slot
    hipEvent_t copied[DFE_NSLOT] = {}, consumed[DFE_NSLOT] = {};
    bool slot_used[DFE_NSLOT] = {};
    int slot_next = 0;
    void *aux = nullptr;              // grow-only side buffer for small per-call planes (the matrix-core matcher's norms): NOT the arena, whose
    size_t aux_bytes = 0;             // carved pointers a nested launcher must not invalidate
    // nn.SpatialContrastiveNormalization's border-correction plane (the estimator of a tensor of ones): a function of the frame size, the
    // plane count and the kernel only -- kept from call to call (filters.hip)
    float *cn_coef = nullptr;
    size_t cn_coef_floats = 0;
    int cn_key[4] = {0, 0, 0, 0};     // H, W, C, k of the plane that is there (k = 0: none)
    float cn_key_kn[33] = {};
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
    // attach mode (a scope around ONE launch): the pair of events rides on the kernel's own dispatch packet (hipExtLaunchKernelGGL(...,
    // prof.a, prof.b, 0, ...)) instead of being recorded on the stream in front of and behind it.  A recorded event is a barrier packet:
    // the two of them cost the VGA step 3.9 us (0.2472 against 0.2433 ms per step with and without the brackets) -- of a measurement
    // that bench.py has to make inside its timed region.  a == b == nullptr: not profiling (or nested), a plain launch.
    hipEvent_t a = nullptr, b = nullptr;
    bool attach;
    explicit DfeProfScope(dfe_ctx *c, bool attach_to_launch = false) : ctx(c), attach(attach_to_launch) {
        if (ctx->profile && ctx->prof_depth++ == 0) {
            hipEvent_t e;
            if (attach) {
                if (hipEventCreate(&a) != hipSuccess) a = nullptr;
                if (a && hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); a = b = nullptr; }
            } else if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, ctx->stream); ctx->prof_events.push_back(e); }
        }
    }
    ~DfeProfScope() {
        if (ctx->profile && --ctx->prof_depth == 0) {
            if (attach) {
                if (a && b) { ctx->prof_events.push_back(a); ctx->prof_events.push_back(b); }
            } else if (ctx->prof_events.size() & 1) {
                hipEvent_t e;
                if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, ctx->stream); ctx->prof_events.push_back(e); }
            }
        }
    }
    DfeProfScope(const DfeProfScope &) = delete;
    DfeProfScope &operator=(const DfeProfScope &) = delete;
};

// brackets the launches of one pipeline stage (DFE_STAGE_*) with events when the stage timers are on
struct DfeStageScope {
    static constexpr size_t kMaxStageEvents = 4096;   // event pairs kept until dfe_stage_timers_read collects them
    dfe_ctx *ctx;
    bool rec = false;
    dfe_ctx::StageEvent ev{};
    DfeStageScope(dfe_ctx *c, int stage) : ctx(c) {
        if (ctx->stage_timers && ctx->stage_depth++ == 0 && ctx->stage_events.size() < kMaxStageEvents) {   // (unread regions beyond the cap are dropped)
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
// arena of at least `bytes`.  plain = false: physically contiguous memory if the driver has it (the volume sweeps' arena); plain = true: a
// plain hipMalloc, for the paths whose convolutions write many feature planes side by side (see dfe_scratch in dfe_ctx.hip for both measurements)
int dfe_scratch(dfe_ctx *ctx, size_t bytes, void **out, bool plain = false);
// one layer of a filter stack (filters.hip): in [nIn][H][W] -> out [nOut][H-kH+1][W-kW+1], nn.Tanh fused behind it when
// L.tanh_after (the same tanhf as dfe_tanh_f32: bit-identical to the two separate calls)
int dfe_filter_layer_forward(dfe_ctx *ctx, const float *in, const dfe_filter_layer &L, int H, int W, float *out);
// the same layer position of n independent inputs (both frames of every pyramid scale) in ONE launch where a batched kernel exists
int dfe_filter_layer_forward_batch(dfe_ctx *ctx, int n, const float *const *in, const dfe_filter_layer *const *L, const int *H, const int *W,
                                   float *const *out);
int dfe_filter_layer_forward_batch_view(dfe_ctx *ctx, int n, const float *const *in, const dfe_filter_layer *const *L, const int *H, const int *W, const int *in_pitch,
                                        const long long *in_plane, float *const *out, bool *done);
// the same layer of up to two inputs (views allowed) as an implicit GEMM on the matrix cores, weights resident in LDS (conv_mfma.hip):
// fused multiply-adds in the reference's (input plane, ky, kx) order -- results differ from the exact kernels by that fusing only
// nrm[e] (or NULL): the kernel also leaves the per-pixel squared norm of its output over the planes there ([Ho][Wo])
int dfe_conv_mfma_res_batch(dfe_ctx *ctx, int n, const float *const *in, const int *H, const int *W, const int *in_pitch, const long long *in_plane,
                            const dfe_filter_layer &L, float *const *out, bool *handled, float *const *nrm = nullptr);
// nn.SpatialContrastiveNormalization with caller-provided scratch ((C + 3) * H * W floats): for the one-call pipelines (filters.hip)
int dfe_contrastive_normalization_run(dfe_ctx *ctx, const float *in, int C, int H, int W, const float *kernel_host, int k, float threshold,
                                      float thresval, float *scratch, float *out);
// ... of two frames of one size in the same two launches; out0 only the crop window cw x ch at (cx, cy) when cw > 0 (scratch: 4 * H * W floats)
int dfe_contrastive_normalization_run2(dfe_ctx *ctx, const float *in0, const float *in1, int C, int H, int W, const float *kernel_host, int k,
                                       float threshold, float thresval, float *scratch, float *out0, float *out1, int cx, int cy, int cw, int ch);
// the raw-patch pyramid on uint8 frames, converted inside its preparation kernels (multiscale.hip; f16_scale 0 = fp32 volumes)
int dfe_multiscale_flow_pair_bytes(dfe_ctx *ctx, const uint8_t *I0, const uint8_t *I1, int C, int H, int W, int k, int maxh, int maxw,
                                   const int *ratios, int nratios, float u8_scale, float f16_scale, float *flow, int64_t *idx);
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
                           // 8 x (minimum, first index as int bits) | 8 x centre cost | 8 x 0 | lead cells -- whole lines, written by ONE store of one wave,
                           // a block sweeping down its column writes consecutive lines.  (The planes took four partial-line stores per
                           // PIXEL and row step -- 8 B + 8 B + 64 B + 4 B, lines shared with neighbouring blocks on other XCDs -- and made
                           // the fused 1080p kernel take 1.78 .. 2.21 ms depending on the process; a 128-B record per pixel, 251 MB at
                           // 1080p, cost 0.8 ms: DESIGN section 5.)  Behind the first line: [pixel][DFE_REC_NLEAD] the pixels' first cells
                           // (extractOutput's input; read back from the volume at first -- that doubled the finalize kernel and, at 1080p,
                           // left volume lines in the memory-side cache that slowed the next launch's stores by 10 %).
    int rec_rows;          // output rows of the pair (the record's row pitch)
};
#ifndef DFE_REC_NLEAD
#define DFE_REC_NLEAD 8    // a pixel's first cells kept in its tile row's record (0: none, extractOutput reads them from the volume)
#endif
#define DFE_REC (32 + 8 * DFE_REC_NLEAD)   // floats per tile-row record (1 or 3 whole 128-B lines)
#define DFE_REC_CENTRE 16  // (entries 0..15: (minimum, index) of the 8 pixels; 16..23: their centre costs; 24..31: 0)
#define DFE_REC_LEAD 32    // (entries 32..: [pixel][DFE_REC_NLEAD] the pixels' first cells)
// ---- the finalize of a pixel from its tile row's record: flow_finalize_kernel's record path (postops.hip), kept here next to the record layout it reads
// (round 4 also ran it at the end of the fused sweep: no gain, ssd_cost_volume.hip) ----
// replaces: radial/radial_opticalflow_groundtruth.lua:87-105 (min(3), tie-break, decode, extractOutput)
struct TailOut {
    long long *idx;      // [P] or null
    float *best;         // [P] or null
    float *fy, *fx;      // decoded displacement, written at (y+pad_t)*pitch + x+pad_l  (pad-back :108), or null
    float *scores;       // extractOutput score, same addressing as fy/fx when padded != 0, else [P]
    long long *imaxs;    // [P] or null (goes with scores)
    int Wo;              // pixels per volume row
    int pitch, pad_t, pad_l;   // full-frame addressing for fy/fx/(scores if padded)
    int padded;          // scores addressed full-frame (1) or [P] (0)
    long long p_off;     // pixel offset of this band inside the [P] outputs
    int row_off;         // output-row offset of this band
    // frame mode (flow_finalize_kernel, one band only): the threads cover the whole H x W frame -- interior pixels run the
    // pipeline's tail and the flow -> depth formula, border pixels are zeroed -- so the pair step needs no third launch
    int frame_H, frame_W;     // 0 = off
    float *depth, *conf;      // [H][W] or null
    float mw, mh, infty;      // focus of expansion, depth clamp (test_opticalflow.lua:143-216)
};

// flow -> depth of one pixel (i, j) with displacement (dy, dx): the quirk-preserving cartesian formula of
// test_opticalflow.lua:143-216 (same arithmetic as flow_to_depth_cartesian_kernel)
__device__ __forceinline__ void pair_depth_px(int i, int j, float dy, float dx, float mw, float mh, float infty, float *r_out, float *c_out) {
    const float py = (float)i - mh, px = (float)j - mw;
    const float pn = (float)sqrt((double)(px * px + py * py));
    const float dn = (float)sqrt((double)(dx * dx + dy * dy));
    float r = 0.f, c = 0.f;
    if (dn >= 0.2f) {
        const float q = pn / dn;
        r = q < infty ? q : infty;
        if (px * dx + dy * dy > 0.125f) c = 1.0f;   // test_opticalflow.lua:181 (sic)
    } else {
        c = 1.0f;
        r = infty;
    }
    *r_out = r;
    *c_out = c;
}


// A6: the record's (minimum, first index), centre override.  A9: decode.  A7: extractOutput over the pixel's first DFE_REC_NLEAD cells
// (in the record), walking on through the volume itself only if fewer than M of them pass the threshold (extract_output.cpp:99-112 stops
// at M as well).  p: pixel index inside the band (row-major over Wo); (fi, fj): its frame position (frame mode).
template <int M>
__device__ __forceinline__ void dfe_finalize_rec_pixel(const float *__restrict__ rec, int rec_rows, const float *__restrict__ vol, long long p, int N,
                                                       int hWin, int wWin, int middle, double threshold, const TailOut &o, int fi, int fj,
                                                       int iy = -1, int ix = -1) {
    // (iy, ix): the pixel's row / column inside the band where the caller has them (frame mode) -- else from p, as a 32-bit division
    // (the 64-bit quotient and remainder of the first version were a hundred instructions of a kernel that has few others)
    const long long pg = o.p_off + p;
    const int yb = iy >= 0 ? iy : (int)((unsigned)p / (unsigned)o.Wo), x = ix >= 0 ? ix : (int)((unsigned)p - (unsigned)yb * (unsigned)o.Wo);
    const int y = yb + o.row_off;
    const int ncols = (o.Wo + 7) >> 3;
    const int g = min(x >> 3, ncols - 1), xb = g == ncols - 1 ? o.Wo - 8 : g << 3;   // (the last tile column is shifted inwards)
    const float *rp = rec + ((long long)g * rec_rows + y) * DFE_REC;
    // (non-temporal: what is read here is REWRITTEN by the next frame's cost-volume launch -- lines left in the memory-side cache by
    //  these reads made that launch's stores slower: 1080p 2.4 against 1.8 ms)
    // (the pixel's (minimum, index) pair as ONE 8-byte load and its eight lead cells as two 16-byte loads -- the record is 128-B aligned
    //  and both pieces are naturally aligned inside it: four load instructions per pixel instead of eleven)
    typedef float dfe_f2v __attribute__((ext_vector_type(2)));
    typedef float dfe_f4v __attribute__((ext_vector_type(4)));
    float2 b;
    {
        const dfe_f2v bv = __builtin_nontemporal_load(reinterpret_cast<const dfe_f2v *>(rp + 2 * (x - xb)));
        b.x = bv[0]; b.y = bv[1];
    }
    const float cen = __builtin_nontemporal_load(rp + DFE_REC_CENTRE + x - xb);
    long long id = (long long)__float_as_int(b.y) + 1;
    if (middle > 0 && b.x == cen) id = middle;
    if (o.idx) o.idx[pg] = id;
    if (o.best) o.best[pg] = b.x;
    const long long fo = (long long)(y + o.pad_t) * o.pitch + x + o.pad_l;
    const int id0 = (int)id - 1, fl = id0 / wWin;                               // (id <= hWin * wWin: 32-bit)
    const float dyf = (float)(fl - (hWin - 1) / 2), dxf = (float)(id0 - fl * wWin - (wWin - 1) / 2);
    if (o.fy) o.fy[fo] = dyf;
    if (o.fx) o.fx[fo] = dxf;
    if (o.frame_H && o.depth) pair_depth_px(fi, fj, dyf, dxf, o.mw, o.mh, o.infty, &o.depth[fo], &o.conf[fo]);
    if (o.scores) {
        static_assert(DFE_REC_NLEAD == 8, "the record holds a pixel's first 8 cells");
        float hv[M], hi[M];
#pragma unroll
        for (int j = 0; j < M; ++j) { hv[j] = 0.f; hi[j] = 0.f; }
        int n = 0;
        float qq[DFE_REC_NLEAD];
        const float *lv = rp + DFE_REC_LEAD + (x - xb) * DFE_REC_NLEAD;
        {
            const dfe_f4v q0 = __builtin_nontemporal_load(reinterpret_cast<const dfe_f4v *>(lv)), q1 = __builtin_nontemporal_load(reinterpret_cast<const dfe_f4v *>(lv) + 1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { qq[kk] = q0[kk]; qq[4 + kk] = q1[kk]; }
        }
#pragma unroll
        for (int kk = 0; kk < DFE_REC_NLEAD; ++kk) {
            if (kk < N && n < M && (double)qq[kk] > threshold) {
#pragma unroll
                for (int j = 0; j < M; ++j)
                    if (j == n) { hv[j] = qq[kk]; hi[j] = (float)(kk + 1); }
                ++n;
            }
        }
        if (n < M && N > DFE_REC_NLEAD) {   // rare: keep scanning the volume itself
            const float *v = vol + p * N;
            for (int kk = DFE_REC_NLEAD; kk < N && n < M; ++kk) {
                const float t = v[kk];
                if ((double)t > threshold) {
#pragma unroll
                    for (int j = 0; j < M; ++j)
                        if (j == n) { hv[j] = t; hi[j] = (float)(kk + 1); }
                    ++n;
                }
            }
        }
        if (hv[0] > 0) {
            if (M == 4) dfe_sort4(hv, hi); else dfe_sort8(hv, hi);
            if (o.imaxs) o.imaxs[pg] = (long long)hi[0];
#pragma unroll
            for (int j = 1; j < M; ++j) hv[j] += hv[j - 1];
            double acc = 0;
#pragma unroll
            for (int j = 0; j < M; ++j) acc += hv[j];
            o.scores[o.padded ? fo : pg] = (float)acc;
        } else if (o.padded) {
            o.scores[fo] = 0.f;   // pair mode: the caller's buffer is not pre-zeroed (pixels without a hit read 0)
        }
    }
}
// frame mode of dfe_flow_finalize (one band only): finalize also zeroes the frame border and makes depth / confidence
struct DfePairDepth { int H, W; float cx, cy; float *depth, *conf; };
// (rec != nullptr: part / centre / lead are ignored -- minimum / index / centre come from the tile-row records [col group][rec_rows][DFE_REC],
//  the lead cells from the records too (DFE_REC_NLEAD per pixel), further cells -- rarely needed -- from the volume itself)
int dfe_flow_finalize(dfe_ctx *ctx, const float2 *part, const float *centre, const float *lead, int nchunks, long long Ptot,
                      const float *vol, double threshold, int rows, int Wo, int hWin, int wWin, int row_off, int64_t *idx, float *best,
                      float *fy, float *fx, float *scores, int64_t *imaxs, int pitch, int pad_t, int pad_l, int scores_padded,
                      const struct DfePairDepth *pd = nullptr, const float *rec = nullptr, int rec_rows = 0);
struct TailOut;
void dfe_make_tailout(TailOut *o, int64_t *idx, float *best, float *fy, float *fx, float *scores, int64_t *imaxs, int Wo, int pitch, int pad_t, int pad_l,
                      int scores_padded, int row_off, const struct DfePairDepth *pd);
int dfe_pair_border_depth(dfe_ctx *ctx, float *flow, float *scores, int H, int W, int pad_t, int pad_l, int Ho, int Wo, float cx,
                          float cy, float *depth, float *conf);
int dfe_feat_matching_fast(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh, int maxw, float *out,
                           bool *handled);
// 16- / 17-wide windows on feature maps: flat tiles, persistent blocks, LDS-DMA staging (feat_matching_flat.hip)
int dfe_feat_matching_flat(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh, int maxw, float *out, bool *handled);
int dfe_feat_matching_flat_argmin(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh, int maxw, long long *idx, float *xflow,
                                  float *yflow, bool *handled);
bool dfe_feat_matching_flat_argmin_takes(const dfe_ctx *ctx, int K, int H1, int W1, int maxh, int maxw);
int dfe_feat_matching_flat_strided(dfe_ctx *ctx, const float *in1, int pitch1, long long plane1, const float *in2, int K, int H1, int W1, int maxh, int maxw,
                                   float *out, bool *handled);
// what getModel's tail + processOutput leave per pair (opticalflow_model.lua:201-252): the centre-pasted full-frame flow and confidences,
// optionally the per-pixel class index and extractOutput score over the model's own output region
struct DfeSoftOut {
    int use_threshold;             // 0: processOutput(geometry, out, true, nil);  1: ...(geometry, out, true, threshold)
    float threshold;
    int hFull, wFull;              // geometry.hImg, geometry.wImg
    float *full, *full_conf;       // [2][hFull][wFull] (plane 0 = y), [hFull][wFull]: ZEROED by the caller; the kernel writes the pasted region
    long long *index;              // [H1][W1] or NULL
    float *scores;                 // [H1][W1] or NULL
};
int dfe_feat_matching_flat_soft(dfe_ctx *ctx, const float *in1, int pitch1, long long plane1, const float *in2, int K, int H1, int W1, int maxh, int maxw,
                                const DfeSoftOut *soft, bool *handled);
// the matcher as a banded GEMM on the matrix cores (feat_matching_mfma.hip; option fm_mfma); norms: dfe_feat_matching_mfma_scratch floats
bool dfe_feat_matching_mfma_takes(const dfe_ctx *ctx, int K, int H1, int W1, int maxh, int maxw);
size_t dfe_feat_matching_mfma_scratch(int H1, int W1, int maxh, int maxw);
// norms_ready: `norms` already holds |a|^2 [H1][W1] | |b|^2 [H2][W2] (left there by the convolution that made the features)
int dfe_feat_matching_mfma(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh, int maxw, float *norms, float *out, long long *idx,
                           float *xflow, float *yflow, bool *handled, bool norms_ready = false);
int dfe_aux_scratch(dfe_ctx *ctx, size_t bytes, void **out);   // the ctx's side buffer, grown to at least `bytes`
bool dfe_feat_matching_win64_ok(const dfe_ctx *ctx, int K, int maxh, int maxw);   // the ctx / window conditions of the launcher below
int dfe_feat_matching_win64_batch(dfe_ctx *ctx, int n, const float *const *in1, const float *const *in2, int K, const int *H1, const int *W1, int maxh,
                                  int maxw, float *const *out, float f16_scale, bool *handled, const struct CvFineArgs *fine = nullptr);
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

// Reductions over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), partners at distance 8, 4, 2, 1: the soft-max of windows of more
// than 64 cells gives a pixel to 16 lanes (softmin_body in multiscale.hip and the feature matcher's soft-max epilogue share this order,
// so the one-call single-scale model equals the staged modules bit for bit)
__device__ __forceinline__ float row16_max_f32(float v) {
#define DFE_STEP(ctrl) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false)))
    DFE_STEP(0x128); DFE_STEP(0x124); DFE_STEP(0x4E); DFE_STEP(0xB1);   // row_ror:8, row_ror:4, quad_perm [2,3,0,1], [1,0,3,2]
#undef DFE_STEP
    return v;
}
__device__ __forceinline__ float row16_sum_f32_ordered(float v) {
#pragma clang fp contract(off)
#define DFE_STEP(ctrl) v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false))
    DFE_STEP(0x128); DFE_STEP(0x124); DFE_STEP(0x4E); DFE_STEP(0xB1);
#undef DFE_STEP
    return v;
}
// (value, index) -> the row's largest value and, among equal values, the smallest index
__device__ __forceinline__ void row16_argmax_first(float &b, int &bi) {
#define DFE_STEP(ctrl)                                                                                        \
    {                                                                                                         \
        const float ob = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), ctrl, 0xf, 0xf, false)); \
        const int oi = __builtin_amdgcn_update_dpp(0, bi, ctrl, 0xf, 0xf, false);                             \
        if (ob > b || (ob == b && oi < bi)) { b = ob; bi = oi; }                                              \
    }
    DFE_STEP(0x128); DFE_STEP(0x124); DFE_STEP(0x4E); DFE_STEP(0xB1);
#undef DFE_STEP
}

// Eight wave reductions at once, "transposed": v[x] is pixel x's value in this lane's cell; the butterfly's first three steps pair
// up REGISTERS as well as lanes (v_permlane32_swap / v_permlane16_swap exchange half-waves / rows between two registers, so one swap
// + one op serves two pixels), halving the live registers at every step, and the last three run on one register.  The result for
// pixel g ends up in lane 8 g.  Partners at distance 32, 16, 8, 4, 2, 1 and the lower lane's value on the left of every +: the sum
// has the association of wave_sum_f32_ordered / px_softmin64 bit for bit.  18 VALU operations instead of 8 x 12.
// OP 0: fp32 sum; 1 / 2: minimum / maximum of NON-NEGATIVE floats, taken on their bit patterns as integers (the same order, and no
// canonicalising v_max x, x, x in front of every operand the way fminf / fmaxf compile).
template <int OP>
__device__ __forceinline__ int wave_reduce8_op(int a, int b) {
#pragma clang fp contract(off)
    return OP == 0 ? __float_as_int(__int_as_float(a) + __int_as_float(b)) : OP == 1 ? min(a, b) : max(a, b);
}
template <int OP>
__device__ __forceinline__ int wave_reduce8_transposed(const float (&v)[8], int lane) {
    int r1[4], r2[2];
#pragma unroll
    for (int x = 0; x < 4; ++x) {        // lanes < 32: pixel x, lanes >= 32: pixel x + 4
        const auto q = __builtin_amdgcn_permlane32_swap(__float_as_int(v[x]), __float_as_int(v[x + 4]), false, false);
        r1[x] = wave_reduce8_op<OP>((int)q[0], (int)q[1]);
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {        // rows 0..3: pixels x, x + 2, x + 4, x + 6
        const auto q = __builtin_amdgcn_permlane16_swap(r1[x], r1[x + 2], false, false);
        r2[x] = wave_reduce8_op<OP>((int)q[0], (int)q[1]);
    }
    const bool up = (lane & 8) != 0;     // from here on lane L works for pixel L >> 3
    const int keep = up ? r2[1] : r2[0], send = up ? r2[0] : r2[1];
    int r = wave_reduce8_op<OP>(keep, __builtin_amdgcn_update_dpp(0, send, 0x128, 0xf, 0xf, true));                  // row_ror:8
    r = wave_reduce8_op<OP>(r, __builtin_amdgcn_update_dpp(0, r, 0x104, 0xf, 0xf, true));                           // row_shl:4 (lane j reads lane j + 4)
    r = wave_reduce8_op<OP>(r, __builtin_amdgcn_update_dpp(0, r, 0x4E, 0xf, 0xf, true));                            // quad_perm [2,3,0,1]
    r = wave_reduce8_op<OP>(r, __builtin_amdgcn_update_dpp(0, r, 0xB1, 0xf, 0xf, true));                            // quad_perm [1,0,3,2]
    return r;                             // lane 8 g: pixel g (other lanes: partial results)
}

// The finest scale of the multiscale matcher, consumed where it is produced: a task row is 8 pixels x the 64 cells of their 8 x 8
// windows, lane <-> cell.  Per pixel: soft-min over the wave (wave minimum of the costs, exponential, wave sum in the association
// order of every other soft-min on the device, e * (1 / sum)), cascade add of the parent pixel's window (cell (a, b) reads the
// parent's cell (2 + a/2, 2 + b/2): one ds_bpermute of the parent value every lane holds for its own cell), arg-max over the 64
// classes of this scale (wave maximum, lowest lane attaining it) against the coarser chain's running best (this scale wins ties:
// its class ids are smaller), centre override, decode -- the operations of cascade_px_kernel<FINEST> in the lane <-> cell form, on
// the same values in the same order: bit-identical results, and the scale-1 volume (84 % of the pyramid's bytes) is never written
// or read.  The three reductions run for the 8 pixels together (wave_reduce8_transposed); lane 8 g finishes pixel g and stores it.
//   (-c) - max(-c) == min(c) - c bit for bit; costs are sums of squares (>= +0) and the cascaded values sums of probabilities, so the
//   integer order of the bit patterns is the float order (frames with NaN / Inf give garbage on either path, not the same garbage).
//   Centre override (bv == centre value): the centre is one of the 64 cells, so centre <= fv; if the coarser chain's best wins
//   (pbv > fv) it is larger than the centre, otherwise bv = fv and the test is "the centre cell attains the maximum" = its bit in
//   the ballot the arg-max needs anyway.
//   MID (a scale > 1 with a coarser one above it, cascade_px_kernel<false>): the cascaded window is stored for the next finer scale, and
//   the arg-max runs over the 48 ring cells in CLASS order (top two rows, left 4 x 2, right 4 x 2, bottom two rows): cells outside
//   the ring take the most negative integer before the maximum; among the cells that attain it the class order is "first non-empty
//   group, lowest cell in it" -- scalar arithmetic on the ballot.
template <int TX, bool F16, bool MID>
__device__ __forceinline__ void fine_epilogue(const float (&vrow)[TX], int lane, int y, int xt, int Wo, const CvFineArgs &fa) {
#pragma clang fp contract(off)
    static_assert(TX == 8, "8 fine pixels = 4 parent pixels");
    const int a = lane >> 3, b = lane & 7;
    const int gsrc = (((2 + (a >> 1)) << 3) + 2 + (b >> 1)) << 2;          // byte address for ds_bpermute: the parent cell this cell adds
    const bool has_parent = fa.pcasc != nullptr;                            // (launch-uniform)
    float par[4];
    float2 pb = make_float2(0.f, 0.f);
    if (has_parent) {
        const long long pp = (long long)(y >> 1) * (Wo >> 1) + (xt >> 1);
        const float *pc = fa.pcasc + pp * 64 + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) par[j] = pc[j * 64];
        pb = fa.pbest[pp + (lane >> 4)];                                    // lane 8 g: the running best of pixel g's parent
    }
    float v[TX];
#pragma unroll
    for (int x = 0; x < TX; ++x) v[x] = F16 ? (float)(_Float16)(vrow[x] * fa.f16_scale) * fa.f16_inv : vrow[x];   // what a stored fp16 volume would hold
    const int mn = wave_reduce8_transposed<1>(v, lane);
    int bc[TX];
#define DFE_BCAST8(src)                                                                                                          \
    _Pragma("unroll") for (int x = 0; x < TX; ++x) bc[x] = __builtin_amdgcn_readlane(src, 8 * x);                                \
    asm volatile("" : "+s"(bc[0]), "+s"(bc[1]), "+s"(bc[2]), "+s"(bc[3]), "+s"(bc[4]), "+s"(bc[5]), "+s"(bc[6]), "+s"(bc[7]))   // (all eight read before the first use: no wait states between a v_readlane and its consumer)
    DFE_BCAST8(mn);
#pragma unroll
    for (int x = 0; x < TX; ++x) v[x] = dfe_exp_nonpos(__int_as_float(bc[x]) - v[x]);
    const int rs = __float_as_int(1.0f / __int_as_float(wave_reduce8_transposed<0>(v, lane)));
    DFE_BCAST8(rs);
#undef DFE_BCAST8
#pragma unroll
    for (int x = 0; x < TX; ++x) v[x] = v[x] * __int_as_float(bc[x]);
    if (has_parent) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g = __int_as_float(__builtin_amdgcn_ds_bpermute(gsrc, __float_as_int(par[j])));
            v[2 * j] = v[2 * j] + g;
            v[2 * j + 1] = v[2 * j + 1] + g;
        }
    }
    if constexpr (MID) {
        float *cq = fa.casc + ((long long)y * Wo + xt) * 64 + lane;
        const bool ring = !(a >= 2 && a <= 5 && b >= 2 && b <= 5);
#pragma unroll
        for (int x = 0; x < TX; ++x) {
            cq[x * 64] = v[x];
            v[x] = ring ? v[x] : __int_as_float(0x80000000);
        }
    }
    const int fvp = wave_reduce8_transposed<2>(v, lane);
    const int mbit = (fa.middle - 1) & 63;
    unsigned long long codes = 0;                                           // byte x: pixel x's first maximal cell | centre-is-maximal << 6
#pragma unroll
    for (int x = 0; x < TX; ++x) {
        const unsigned long long hit = __builtin_amdgcn_ballot_w64(__float_as_int(v[x]) == __builtin_amdgcn_readlane(fvp, 8 * x));
        unsigned long long code;
        if constexpr (MID) {                                                // (byte x: the class rank 0..47 of pixel x's first maximal ring cell)
            const unsigned long long top = hit & 0xffffull, left = hit & 0x0000030303030000ull, right = hit & 0x0000c0c0c0c00000ull;
            const int cell = __builtin_ctzll(top ? top : left ? left : right ? right : hit);
            const int side = 16 + ((cell >> 3) - 2) * 2 + (cell & 7);      // left columns 0, 1 -> ranks 16..23; right columns 6, 7 -> 24..31
            code = (unsigned long long)(cell < 16 ? cell : cell >= 48 ? cell - 16 : (cell & 7) < 2 ? side : side + 2);
        } else {
            code = (unsigned long long)__builtin_ctzll(hit) | (((hit >> mbit) & 1ull) << 6);
        }
        codes |= code << (8 * x);
    }
    if constexpr (MID) {
        if ((lane & 7) == 0) {
            const int g = lane >> 3;
            float bv = __int_as_float(fvp);
            int bi = ((int)(codes >> (8 * g)) & 0xff) + fa.cls_base;
            if (has_parent && !(bv >= pb.x)) { bv = pb.x; bi = __float_as_int(pb.y); }      // the scale wins ties against the coarser chain
            fa.best[(long long)y * Wo + xt + g] = make_float2(bv, __int_as_float(bi));
        }
        return;
    }
    if ((lane & 7) == 0) {
        const int g = lane >> 3;
        const int code = (int)(codes >> (8 * g)) & 0xff;
        int bi = code & 63;
        bool centre = (code & 64) != 0;
        if (has_parent && !(__int_as_float(fvp) >= pb.x)) { bi = __float_as_int(pb.y); centre = false; }   // the scale wins ties against the coarser chain
        int id = bi + 1;
        if (fa.middle > 0 && centre) id = fa.middle;
        const long long p = (long long)y * Wo + xt + g;
        if (fa.idx) fa.idx[p] = id;
        if (fa.fy) {
            const int d = fa.dec[id - 1];
            fa.fy[p] = (float)(d >> 16);
            fa.fx[p] = (float)(short)(d & 0xffff);
        }
    }
}

#endif  // __HIPCC__

