/*
 * dfe.h -- C ABI of libdfe.so: the MI355X (gfx950) implementation of the dense
 * patch-correlation flow->depth hot path of MichaelMathieu/depth-estimation.
 *
 * This is the drop-in boundary.  Each entry point replaces one operator the
 * reference reaches through Torch7's nn.Module protocol or through its two
 * native Lua C modules; the replaced reference interface is cited as
 * "replaces: file:line" (paths relative to the reference repository).  A
 * LuaJIT-FFI caller binds exactly these prototypes.
 *
 * Conventions
 *  - plain C types only; every tensor is a dense row-major buffer, sizes in elements;
 *  - all tensor pointers are DEVICE pointers (hipMalloc / dfe_malloc / any allocator of the
 *    same HIP runtime, e.g. torch).  dfe_memcpy_* stage host data for callers without one;
 *  - the caller allocates inputs AND outputs (the reference's ops also write into
 *    caller-allocated tensors, extract_output.cpp:63-81); the library owns only ctx scratch;
 *  - class ids are 1-based int64 exactly as the reference's LongTensors;
 *  - every function returns DFE_OK or a negative DFE_E_*; dfe_last_error(ctx) gives the text
 *    (the Lua skin turns that into error(), matching CascadingAddTable.lua:111,115,123);
 *  - work is enqueued on the ctx's HIP stream and NOT synchronised unless stated;
 *  - a ctx is single-threaded; use one ctx per GPU / host thread;
 *  - there is no CPU fallback: without a gfx950 device dfe_ctx_create fails with DFE_E_HIP.
 */
#ifndef DFE_H
#define DFE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DFE_OK 0
#define DFE_E_ARG (-1)
#define DFE_E_SHAPE (-2)
#define DFE_E_ALLOC (-3)
#define DFE_E_HIP (-4)
#define DFE_E_UNSUPPORTED (-5)

#define DFE_MAX_RATIOS 10 /* x2yxMulti2.c:1 N_MAX_RATIOS */

typedef struct dfe_ctx dfe_ctx;

/* ---- context, memory --------------------------------------------------- */
int dfe_version(void);
/* revision tag of the cost-volume kernels (profiles/traffic_*.json records the one its HBM counters were measured with) */
const char *dfe_kernel_revision(void);
/* own_stream != 0: the ctx creates (and owns) a private non-blocking stream, `stream` is ignored;
 * own_stream == 0: work is enqueued on the caller's hipStream_t `stream` (NULL = the HIP default
 * stream, which is what torch's default stream is) so it is ordered with the caller's other work */
int dfe_ctx_create(int device, void *stream, int own_stream, dfe_ctx **out);
void dfe_ctx_destroy(dfe_ctx *ctx);
const char *dfe_last_error(const dfe_ctx *ctx); /* ctx may be NULL: last creation error */
int dfe_ctx_synchronize(dfe_ctx *ctx);
void *dfe_ctx_stream(dfe_ctx *ctx);
int dfe_malloc(dfe_ctx *ctx, size_t bytes, void **dptr);
int dfe_free(dfe_ctx *ctx, void *dptr);
int dfe_memcpy_h2d(dfe_ctx *ctx, void *dst, const void *src, size_t bytes); /* synchronous */
int dfe_memcpy_d2h(dfe_ctx *ctx, void *dst, const void *src, size_t bytes); /* synchronous */
/* Pins a host range the caller keeps passing to dfe_memcpy_h2d / _d2h (the storage of a Torch tensor that is reused from
 * frame to frame): copies from / to it then run as direct DMA instead of through the runtime's pageable staging.
 * Registering a range twice, or unregistering an unknown one, is not an error.  Unregister before the memory is freed. */
int dfe_host_register(dfe_ctx *ctx, void *ptr, size_t bytes);
int dfe_host_unregister(dfe_ctx *ctx, void *ptr);
/* Library-owned pinned host memory (hipHostMalloc): the bounce buffer a binding stages its tensors through, so that it never
 * has to pin memory it does not own (garbage-collected temporaries, storage that is resized or freed behind its back). */
int dfe_host_alloc(dfe_ctx *ctx, size_t bytes, void **hptr);
int dfe_host_free(dfe_ctx *ctx, void *hptr);
/* cost-volume kernel selection (tuning / tests; also the DFE_CV_MODE environment variable at context creation):
 * 0 = auto (default: the row-image kernel where it applies -- C=3, 7x7 patch, 769..1096 window cells, e.g. 33x33 --
 * else the tiled kernel, else the reference-order kernel), 1 = force the reference-order kernel (bit-identical float
 * summation order to the CPU path), 2 = force the tiled kernel, 3 = force the row-image kernel
 * (2/3: DFE_E_UNSUPPORTED from the op when the shape has no such kernel) */
int dfe_set_cost_volume_kernel(dfe_ctx *ctx, int mode);
/* tile height of the tiled / row-image kernels: 0 = chosen per shape (default); 2..7 = force NQ (a tile sweeps
 * NQ groups of 6 image rows at k = 7: 6 NQ - 6 output rows); 100 + ty (101..164) = row-image tiles of ty output
 * rows (ty a multiple of 6 at k = 7); 1 = force the row-image kernel's column sweep (unfused build only).  For tuning and for testing. */
int dfe_set_cost_volume_tile(dfe_ctx *ctx, int tyq);
/* Behaviour switches of the launchers (tuning and tests; none is needed for correct results -- every choice has a parity test or is
 * a pure scheduling choice).  value >= 0 forces, -1 restores the launcher's own choice per shape.  Keys: "cascade_px", "fine_fuse",
 * "mid_fuse", "fine_nq", "mid_nq", "prep_tiles", "xpose", "xpose_nt", "soft_epilogue", "conv_batch", "conv_nt10", "fm64", "fm_rows",
 * "sweep_ovh", "sweep_blocks", "debug_arena", "fm_flat", "fm_split", "conv_narrow", "conv_mfma", "fm_mfma", "arena_contig", "graphs".  The library reads the environment ONCE, in dfe_ctx_create
 * (DFE_<KEY> variables of the tuning scripts) -- never inside an op, so an op's behaviour depends on its ctx only.
 * Two keys trade the exact arithmetic for the matrix cores, both OFF unless set to 1: "conv_mfma" (the one-call models' filter layers as
 * implicit GEMMs, v_mfma_f32_16x16x4_f32: the reference's (input plane, ky, kx) order with FUSED multiply-adds, <= 1e-5 relative to
 * sum |terms|) and "fm_mfma" (nn.SpatialMatching as a banded GEMM, |a|^2 + |b|^2 - 2 a.b: costs within 1e-5 |c| + 1e-6 max |c| of the
 * exact k-ordered sums, arg-min indices equal except where two costs lie within that band).
 * replaces: the option tables the reference's drivers pass down (opticalflow.lua:138-198 `geometry`), for the switches that have no
 * counterpart there.  Unknown key: DFE_E_ARG. */
int dfe_set_option(dfe_ctx *ctx, const char *key, int value);
int dfe_get_option(dfe_ctx *ctx, const char *key, int *value);
/* name of the kernel the last cost-volume call launched (static string) */
const char *dfe_last_kernel(const dfe_ctx *ctx);

/* upper bound for the ctx scratch arena that holds a materialised cost volume inside the one-call
 * pipelines (default 16 GiB); larger volumes are processed in bands of output rows */
int dfe_set_scratch_limit(dfe_ctx *ctx, size_t bytes);
/* device memory for a caller-owned cost volume (the `out` of dfe_ssd_cost_volume_f32 and friends): physically contiguous where the driver
 * grants it (hipExtMallocWithFlags + hipDeviceMallocContiguous), a plain hipMalloc otherwise; *contiguous (may be NULL) says which.  The
 * sweeps write a volume at up to 10 % more bytes per second into contiguous memory than into an unlucky plain allocation (the ctx's own arena is
 * allocated the same way, option "arena_contig").  Any device pointer works as `out`; this is the allocation that is fastest to fill.
 * replaces: torch.Tensor():resize() of the matcher's output on the reference side (SpatialMatching.lua:24), for callers that want the
 * placement.  dfe_device_free releases it (NULL is allowed). */
int dfe_device_alloc(dfe_ctx *ctx, size_t bytes, void **ptr, int *contiguous);
int dfe_device_free(dfe_ctx *ctx, void *ptr);
/* per-launch HIP-event timing of the cost-volume kernel on the ctx stream (bench.py's roofline):
 * enable, run, then read the summed kernel time and launch count (read synchronises and resets) */
int dfe_profile_enable(dfe_ctx *ctx, int on);
int dfe_profile_read(dfe_ctx *ctx, double *total_ms, int *launches);
/* the same, and the first `cap` launches' durations one by one (each_ms[cap], in launch order): where inside a timed region the time went
 * (bench.py reports minimum / median / maximum next to the average) */
int dfe_profile_read_each(dfe_ctx *ctx, double *total_ms, int *launches, float *each_ms, int cap);

/* Stage timers with the reference's names -- `load / filter / match / extract` printed per frame by the dense driver
 * (depth_estimation_opticalflow.lua:44-47,112-117,144-148): HIP events around the launches of each stage inside the
 * one-call pipelines and the staging copies.  load = dfe_memcpy_h2d (the driver's image load + upload); filter = down-sampling,
 * padding, polar warps and the learned filter stacks (its `filter:forward`); match = cost volumes, soft-min and cascade
 * (`model:forward`); extract = arg-best / extractOutput / decode / border / depth passes (`processOutput`; where the arg-best
 * is fused into the matching kernel -- the multiscale cascade, the fused cost-volume build -- that part counts as match).
 * enable, run, read: ms[s] = summed GPU time of stage s, launches[s] = number of bracketed regions; read synchronises and resets. */
#define DFE_STAGE_LOAD 0
#define DFE_STAGE_FILTER 1
#define DFE_STAGE_MATCH 2
#define DFE_STAGE_EXTRACT 3
#define DFE_NSTAGES 4
int dfe_stage_timers_enable(dfe_ctx *ctx, int on);
int dfe_stage_timers_read(dfe_ctx *ctx, double *ms /* [DFE_NSTAGES] */, int *regions /* [DFE_NSTAGES] */);

/* ---- A0+A1: dense SSD cost volume from raw frames ----------------------- */
/* replaces: unfold + SpatialPadding crop + nn.SpatialMatching(hWin,wWin,false):forward
 *   radial/radial_opticalflow_groundtruth.lua:79-84 (= version2/groundtruth.lua:77-82),
 *   the raw-patch (identity filter) case of opticalflow_model.lua:81-99.
 * I0,I1 [C][H][W] f32.  out [Ho][Wo][hWin][wWin] f32,
 *   Ho = H-kh+1-hWin+1, Wo = W-kw+1-wWin+1,
 *   out[y][x][dy][dx] = sum_{c,i,j} (I0[c][y+oy+i][x+ox+j] - I1[c][y+dy+i][x+dx+j])^2,
 *   oy = floor((hWin-1)/2), ox = floor((wWin-1)/2). */
int dfe_ssd_cost_volume_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W,
                            int kh, int kw, int hWin, int wWin, float *out);

/* ---- A0+A1 with an fp16 volume (BASELINE configs[4]; SURVEY 7 "fp16 cost volume", 8(c) numeric contract) ------ */
/* Same sums as dfe_ssd_cost_volume_f32 (fp32 accumulation); the volume is stored as IEEE half, out[..] =
 * half(cost * scale), round to nearest even.  Raw SSD overflows half (147 * 255^2 = 9.56e6 > 65504), hence the scale:
 * 2^-8 for uint8-valued frames (max 37 344), 1 for frames in [0, 1].  out [Ho][Wo][hWin][wWin] of uint16 (half bits).
 * Stored values are within rel 2^-11 of the fp32 volume (bit-exact against half(oracle * scale) on integer-valued frames). */
int dfe_ssd_cost_volume_f16(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int kh, int kw, int hWin,
                            int wWin, float scale, void *out);
/* dfe_flow_depth_pair_f32 with the volume materialised as fp16: arg-min (+ centre tie-break) is taken on the fp32 sums
 * BEFORE the down-convert, so idx / best / flow / depth are identical to the fp32 path's.  idx, best [Ho][Wo] (may be NULL),
 * flow [2][H][W] centre-pasted, depth / depth_conf [H][W] (both or neither).  No extractOutput scores (their rare
 * fall-back reads the volume).  Shapes: C in {1,3}, 7x7 patch, 769..1096 window cells (else DFE_E_UNSUPPORTED). */
int dfe_flow_depth_pair_f16(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int k, int hWin, int wWin,
                            float foe_x, float foe_y, float scale, int64_t *idx, float *best, float *flow, float *depth,
                            float *depth_conf);

/* ---- A1: nn.SpatialMatching(maxh,maxw,false):updateOutput on feature maps -- */
/* replaces: nnx SpatialMatching call sites opticalflow_model_multiscale.lua:216,
 *   opticalflow_model.lua:93, version2/network.lua:30, tests/time_matching.lua:18.
 * in1 [K][H1][W1], in2 [K][H1+maxh-1][W1+maxw-1] -> out [H1][W1][maxh][maxw]. */
int dfe_spatial_matching_f32(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1,
                             int W1, int maxh, int maxw, float *out);

/* ---- A1r: nn.SpatialRadialMatching(hWin):updateOutput --------------------- */
/* replaces: radial/radial_opticalflow_network.lua:33,71-72.
 * in1 [K][H1][W], in2 [K][H1+hWin-1][W] -> out [H1][W][hWin]. */
int dfe_radial_matching_f32(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1,
                            int W, int hWin, float *out);

/* ---- A1r + arg-min, and the radial path in one call (BASELINE configs[2]) ------------------ */
/* replaces: SpatialRadialMatching(hWin):forward followed by `_, idx = output:min(3); idx:add(-1)`
 *   (radial/train_radial_opticalflow.lua:161-168, radial/test_radial_opticalflow.lua:204-207).  in1 [K][in1_plane_rows][W]
 *   (only the first H1 rows of a plane are read: the previous frame cropped by hWin-1 rows, network.lua:59;
 *   0 = H1), in2 [K][H1+hWin-1][W]; volume [H1][W][hWin] or NULL; flow [H1][W] = first-minimum index - 1 as float
 *   (TH min keeps the first minimum), its last row zeroed when zero_last_row (train_radial:178-180: the trainer's display
 *   code does that; the inference script test_radial:204-207 does not).
 *   hWin in {8, 12, 15, 16}. */
int dfe_radial_match_argmin_f32(dfe_ctx *ctx, const float *in1, int in1_plane_rows, const float *in2, int K, int H1, int W,
                                int hWin, float *volume, float *flow, int zero_last_row);

/* networkp of the radial scripts (radial/train_radial_opticalflow.lua:83-97) for the default separable filter stack
 * {{C,1,kW1,n1},{n1,kH2,1,n2}} (train_radial:27), optionally with 'tanh' between the two convolutions. */
typedef struct dfe_radial_params {
    int C, hImg, wImg;     /* cartesian frames [C][hImg][wImg] */
    int hInput, wInput;    /* polar image size before the wKernel-1 wrap-around columns */
    int hWin;              /* radial search window */
    int n1, kW1;           /* layer 1: C -> n1 planes, 1 x kW1 (wKernel = kW1) */
    int n2, kH2;           /* layer 2: n1 -> n2 planes, kH2 x 1 (hKernel = kH2) */
    int tanh_between;      /* non-zero: nn.Tanh between the layers */
    float alpha_polar;     /* radial exponent of the polar grid (1 = linear) */
    double kinfty;         /* flow2depth's infinity factor (0.65 in test_radial:225; a Lua number, i.e. a double) */
    int zero_last_row;     /* 0 = radial/test_radial_opticalflow.lua:204-207 as shipped (idx = min(3) - 1, nothing else);
                              non-zero = also `test:sub(h,h,1,w):zero()` of the trainer's display path (train_radial:178-180) */
} dfe_radial_params;
/* rows of the matcher output (hInput - hKernel - hWin + 2) and size of the cartesian flow / depth images
 * (floor(hImg * kOutput) x floor(wImg * kOutput), getP2CMaskOF radial/radial_opticalflow_polar.lua:18-30). */
int dfe_radial_out_shape(const dfe_radial_params *p, int *hMatch, int *hOut, int *wOut);
/* replaces, composed: radial/test_radial_opticalflow.lua:186-225 (= train_radial:161-182 + flow2depth):
 *   getC2PMask + cartesian2polar of both frames -> getTesterNetwork(networkp):forward -> min(3) - 1 -> getP2CMaskOF +
 *   cartesian2polar of the flow -> flow2depth(networkp, flow, e2 * getKOutput(networkp), kinfty).
 *   prev (ego-motion-corrected by the caller) / cur [C][hImg][wImg]; (e2x, e2y) the epipole = focus of expansion in
 *   frame pixels (doubles, like the Lua numbers they replace: they are scaled before the reference rounds them to float); w1 [n1][C][1][kW1], b1 [n1], w2 [n2][n1][kH2][1], b2 [n2] (biases may be NULL).
 *   Outputs (each may be NULL): volume [hMatch][wInput][hWin], polar_flow [hMatch][wInput], cart_flow / depth / conf
 *   [hOut][wOut].  Bit-identical to the staged calls (dfe_polar_grid_c2p_f32, dfe_warp_bilinear_f32,
 *   dfe_spatial_convolution_f32, dfe_radial_matching_f32, ...). */
int dfe_radial_flow_depth_pair_f32(dfe_ctx *ctx, const dfe_radial_params *p, const float *prev, const float *cur, double e2x,
                                   double e2y, const float *w1, const float *b1, const float *w2, const float *b2,
                                   float *volume, float *polar_flow, float *cart_flow, float *depth, float *conf);

/* ---- N2: gradients of the two matchers (training drivers call model:backward through them) ------- */
/* replaces: nn.SpatialMatching:updateGradInput / nn.SpatialRadialMatching:updateGradInput (un-vendored nnx), reached
 *   from radial/train_radial_opticalflow.lua:228-252 and opticalflow.lua:296-338.  gradOut has the forward output's
 *   layout; gradIn1 / gradIn2 the inputs'.  Gather form (no atomics): every input element sums its own terms in
 *   (dy, dx) order. */
int dfe_spatial_matching_backward_f32(dfe_ctx *ctx, const float *in1, const float *in2, const float *gradOut, int K,
                                      int H1, int W1, int maxh, int maxw, float *gradIn1, float *gradIn2);
int dfe_radial_matching_backward_f32(dfe_ctx *ctx, const float *in1, const float *in2, const float *gradOut, int K,
                                     int H1, int W, int hWin, float *gradIn1, float *gradIn2);

/* ---- A6: arg-min / arg-max with the centre tie-break ---------------------- */
/* replaces: output:min(3) + centre override radial/radial_opticalflow_groundtruth.lua:88-94;
 *   input:max(3) + override in getOutputConfidences opticalflow_model.lua:153-161.
 * vol [P][N] -> idx [P] (1-based, first extremum wins, centre wins exact ties with the best),
 *   best [P] (may be NULL).  middle <= 0 disables the override. */
int dfe_argbest_center(dfe_ctx *ctx, const float *vol, int64_t P, int N, int middle, int take_max,
                       int64_t *idx, float *best);

/* ---- A7 / A8: extractoutput.* --------------------------------------------- */
/* replaces: extractoutput.extractOutput(input, scores, threshold, imaxs)
 *   extract_output.cpp:63-155 (Lua registration :357-366); same argument order.
 * input [H][W][N] f32; imaxs [H][W] int64, scores [H][W] f32.  Pixels with no value
 * above threshold are left untouched, as in the reference. */
int dfe_extract_output(dfe_ctx *ctx, const float *input, int H, int W, int N, float *scores,
                       double threshold, int64_t *imaxs);
/* replaces: extractoutput.extractOutputMarginalized(input, threshold, threshold_acc, ret, retgd)
 *   extract_output.cpp:157-255.  retgd is zeroed first (:166). */
int dfe_extract_output_marginalized(dfe_ctx *ctx, const float *input, int H, int W, int N,
                                    double threshold, double threshold_acc, int64_t *ret,
                                    int64_t *retgd);

/* ---- A9 / A10: class id -> displacement ----------------------------------- */
/* replaces: x2yx + centered2onebased opticalflow_model.lua:16-34,209-213;
 *   radial/radial_opticalflow_groundtruth.lua:97-100. */
int dfe_x2yx(dfe_ctx *ctx, const int64_t *idx, int64_t P, int maxh, int maxw, int64_t *y,
             int64_t *x);
/* replaces: x2yxMulti2(geometry, LongTensor) opticalflow_model_multiscale.lua:72-81 (body
 *   x2yxMulti2.c:1-95).  compat_c = 0: the Lua scalar semantics x2yxMultiNumber :83-132 that the
 *   reference's round-trip test pins (tests/test_multiscale.lua:57-80); compat_c = 1: the shipped C
 *   body bug for bug (ids it never matches leave y/x untouched). ratios are the Lua table values
 *   ratios[1..n]. Returns DFE_E_ARG if an id is outside 1..nclasses (compat_c = 0 only; the device
 *   flag is read back, so this call synchronises). */
int dfe_x2yx_multi(dfe_ctx *ctx, int maxh, int maxw, const int *ratios, int nratios,
                   const int64_t *idx, int64_t P, int64_t *y, int64_t *x, int compat_c);
/* host-side scalar codec (no device work): yx2xMulti :10-52, x2yxMultiNumber :83-132 */
int64_t dfe_yx2x_multi(int maxh, int maxw, const int *ratios, int nratios, double y, double x);
int dfe_x2yx_multi_number(int maxh, int maxw, const int *ratios, int nratios, int64_t id,
                          int64_t *y, int64_t *x);
int64_t dfe_multi_nclasses(int maxh, int maxw, const int *ratios, int nratios);

/* ---- dense single-scale flow in one call (A0+A1+A6+A7+A9; the volume lives only in ctx scratch) --- */
/* replaces: compute_cartesian_groundtruth_cross_correlation radial/radial_opticalflow_groundtruth.lua:66-112
 *   up to (not including) the pad-back :108.
 * Outputs, all [Ho][Wo]: idx = arg-min class with centre tie-break (:88-94), flow_y/flow_x
 *   decoded displacement (:97-100), scores/imaxs = extractOutput(cost, thr) (:105; pixels with
 *   nothing above thr keep the caller's values), best = min cost.  Any output may be NULL. */
int dfe_ssd_flow_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int kh,
                     int kw, int hWin, int wWin, double extract_threshold, int64_t *idx,
                     float *best, float *flow_y, float *flow_x, float *scores, int64_t *imaxs);

/* ---- A6+A7+A9 in one pass over an existing volume (rows x Wo pixels, hWin*wWin cells) ------- */
/* replaces: radial/radial_opticalflow_groundtruth.lua:87-105 (min(3), centre tie-break, decode,
 *   extractOutput) and the pad-back of :108 for the float planes: fy/fx (and scores when
 *   scores_padded != 0) are written at [(row_off+y+pad_t)*pitch + x+pad_l]; idx/best/imaxs (and
 *   scores when scores_padded == 0) at [(row_off+y)*Wo + x].  Any output may be NULL; scores and
 *   imaxs of pixels with nothing above the threshold are left untouched. */
int dfe_flow_tail(dfe_ctx *ctx, const float *vol, int rows, int Wo, int hWin, int wWin,
                  double threshold, int row_off, int64_t *idx, float *best, float *fy, float *fx,
                  float *scores, int64_t *imaxs, int pitch, int pad_t, int pad_l, int scores_padded);

/* ---- A12: flow -> depth ----------------------------------------------------------------- */
/* replaces: the inline-C `radial(geometry, flow, mh, mw)` of test_opticalflow.lua:143-216.
 * flow [2][H][W] (plane 0 = y, plane 1 = x); (cx,cy) = focus of expansion = (mw,mh); infty = W/2.
 * depth = min(|p-c|/|flow|, infty) where |flow| >= 0.2 else infty; conf as shipped uses
 * px*dx + dy*dy (:181); fix_dot != 0 selects px*dx + py*dy. */
int dfe_flow_to_depth_cartesian(dfe_ctx *ctx, const float *flow, int H, int W, float cx, float cy,
                                int fix_dot, float *depth, float *conf);

/* ---- the one-call single-scale pipeline: frames -> flow + confidence + depth ------------- */
/* replaces: the per-frame body of the dense drivers (depth_estimation_opticalflow.lua:113-116 +
 *   test_opticalflow.lua:349-355 flow -> depth) for the single-scale raw-patch matcher.
 * flow [2][H][W], scores [H][W], depth [H][W], depth_conf [H][W]: full-frame, zero outside the
 * centre-pasted Ho x Wo region (opticalflow_model.lua:227-249).  The cost volume is materialised
 * in ctx scratch (bands of output rows within dfe_set_scratch_limit) by dfe_ssd_cost_volume_f32's kernel and consumed by dfe_flow_tail. */
int dfe_flow_depth_pair_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W,
                            int k, int hWin, int wWin, float foe_x, float foe_y,
                            double extract_threshold, float *flow, float *scores, float *depth,
                            float *depth_conf);

/* ---- A2: one pyramid scale on raw frames ------------------------------------------------- */
/* replaces: per-ratio branch of getMultiscalePrefilter + matcher opticalflow_model_multiscale.lua:134-173,196-229
 *   (raw-patch identity filter): box down-sample by r, zero-pad by hPatch2-1 = (maxh-1)+(kh-1) split
 *   floor/ceil, crop frame 0 by maxw-1, SpatialMatching(maxh,maxw).  out [H/r][W/r][maxh][maxw] at native
 *   scale (the x r nearest-neighbour upsampling is applied by the consumer's indexing). */
int dfe_downsample_box_f32(dfe_ctx *ctx, const float *img, int C, int H, int W, int r, float *out);
int dfe_pyramid_scale_volume_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W,
                                 int r, int kh, int kw, int maxh, int maxw, float *out);

/* ---- A3: nn.Minus -> nn.SoftMax over the window cells ------------------------------------ */
/* replaces: cascad_preproc opticalflow_model_multiscale.lua:270-279.  cost, prob [P][N]. */
int dfe_softmin_f32(dfe_ctx *ctx, const float *cost, int64_t P, int N, float *prob);

/* ---- A4: nn.CascadingAddTable:updateOutput ------------------------------------------------ */
/* replaces: CascadingAddTable.lua:108-135.  in[s], out[s]: [P][maxh][maxw] for each of the nratios
 *   scales (arrays of device pointers on the HOST).  out[n-1] = in[n-1];
 *   out[i] = in[i] + replicate(crop(out[i+1])).  DFE_E_SHAPE with the reference's message when ratios
 *   and window size are not compatible (:121-124). */
int dfe_cascading_add_f32(dfe_ctx *ctx, const float *const *in, const int *ratios, int nratios,
                          int64_t P, int maxh, int maxw, float *const *out);

/* ---- A4 + A5 + A6 + A10 fused: cascade -> arg-max -> displacement --------------------------------- */
/* replaces: cascad + middle remover + getOutputConfidences (no threshold) + x2yxMulti of processOutput
 *   (opticalflow_model_multiscale.lua:281-333, opticalflow_model.lua:153-161,205-208) in one pass: the joined
 *   [H][W][nclasses] tensor is never materialised.  prob[s] [H/r_s][W/r_s][maxh][maxw] as for dfe_cascade_ring_f32;
 *   idx [H][W] 1-based class id (first maximum, centre tie-break), best [H][W] its value, flow_y / flow_x [H][W]
 *   the decoded displacement in finest-scale pixels; each output may be NULL (flow_y and flow_x together).
 *   Bit-identical to dfe_cascade_ring_f32 -> dfe_argbest_center(take_max) -> dfe_x2yx_multi. */
int dfe_cascade_flow_f32(dfe_ctx *ctx, const float *const *prob, const int *ratios, int nratios, int H, int W,
                         int maxh, int maxw, int64_t *idx, float *best, float *flow_y, float *flow_x);

/* ---- A2..A6 + A10 in one call: the multiscale matcher of a frame pair ------------------------------ */
/* replaces: getModelMultiscale(geometry, true, false):forward({I0, I1}) + processOutput ('max', no threshold) for
 *   the identity patch filter -- opticalflow_model_multiscale.lua:175-333, opticalflow_model.lua:201-226.
 *   I0, I1 [C][H][W] with H, W multiples of every ratio (the caller pads, :234-248).  Per scale: box down-sample,
 *   zero-pad, k x k raw-patch SSD over a maxh x maxw window, softmin; then the fused cascade / ring / arg-max /
 *   decode.  flow [2][H][W] (plane 0 = y, 1 = x, finest-scale pixels) and / or idx [H][W] (1-based class id). */
int dfe_multiscale_flow_pair_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int k,
                                 int maxh, int maxw, const int *ratios, int nratios, float *flow, int64_t *idx);

/* dfe_multiscale_flow_pair_f32 with every scale's cost volume stored as IEEE half (BASELINE configs[4]: "5-level pyramid, fp16
 * cost volume"; SURVEY 8(d): the 4K 5-level volumes are 1.6 GB in fp32).  The SSD sums are fp32; a volume holds
 * half(cost * scale), round to nearest even, and the cascade works on float(stored) * (1 / scale) in fp32 registers: soft-min,
 * cascade adds and arg-max are fp32 as in the f32 entry.  The result is that of the fp32 chain run on volumes rounded to half
 * precision (11 significant bits) where they are stored.  scale: 1 for frames in [0, 1] (and up to a few units), 2^-8 for
 * uint8-valued frames (147 * 255^2 overflows half).  8 x 8 windows with ratios 1, 2, 4, ... (C = 3, k = 7) write and read real
 * half volumes; any other shape builds fp32 volumes and rounds them in place -- the same values. */
int dfe_multiscale_flow_pair_f16(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int k,
                                 int maxh, int maxw, const int *ratios, int nratios, float scale, float *flow, int64_t *idx);

/* ---- A2..A6 + A10 + A15 in one call: the multiscale matcher with its LEARNED patch filters ------------------------------- */
/* One layer of getFilter(geometry) (opticalflow_model.lua:45-79): nn.SpatialConvolution(nIn, nOut, kW, kH), or
 * nn.SpatialConvolutionMap over a connection table when conn != NULL, followed by nn.Tanh when tanh_after (getFilter puts
 * one behind every layer but the last).  All pointers are DEVICE pointers. */
typedef struct dfe_filter_layer {
    int nIn, nOut, kH, kW;
    const float *weight;   /* [nOut][nIn][kH][kW]; with conn: [nConn][kH][kW] */
    const float *bias;     /* [nOut] or NULL */
    const int32_t *conn;   /* SpatialConvolutionMap: [nConn][2] = (from, to), 1-based int32; NULL = full connection */
    int nConn;
    int tanh_after;
} dfe_filter_layer;
/* replaces: getModelMultiscale(geometry, true, false):forward({I0, I1}) + processOutput ('max', no threshold) with
 *   getFilter(geometry) in front of every scale's nn.SpatialMatching -- opticalflow_model_multiscale.lua:175-333 (the filter
 *   branches :196-211, shared or cloned per scale :219-226), equivalently getMultiscalePrefilter (:134-173) followed by the
 *   prefiltered model: the per-scale computation is the same.  Per scale r: box down-sample by r, zero-pad by hPatch2-1 =
 *   (maxh-1)+(hKernel-1) split floor/ceil (hKernel = sum kH - (nlayers-1), the stack's receptive field: opticalflow.lua:154-189),
 *   frame 0 cropped by maxh-1 / maxw-1 (:198-202), the filter stack on both, SpatialMatching(maxh, maxw) on the K-plane
 *   features, soft-min; then cascade / ring / arg-max / decode as in dfe_multiscale_flow_pair_f32.
 *   layers: HOST array [share_filters ? 1 : nratios][nlayers] (scale-major); share_filters != 0: one stack for every scale
 *   (geometry.share_filters).  I0, I1 [C][H][W], C = layers[0].nIn, H and W multiples of every ratio.
 *   f16_scale != 0: volumes rounded to half precision where they are stored (see dfe_multiscale_flow_pair_f16). */
int dfe_multiscale_flow_pair_filtered_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int maxh, int maxw,
                                          const int *ratios, int nratios, const dfe_filter_layer *layers, int nlayers,
                                          int share_filters, float f16_scale, float *flow, int64_t *idx);

/* ---- the single-scale trained model in one call: A15 + A1 + A3 + A6/A7 + A9 + A11 ------------------------------------------- */
/* replaces: what depth_estimation_opticalflow.lua:66-116 (and depth_estimation_api.lua:164-168, test_opticalflow.lua:347-355) run per
 *   frame pair for a model that is not multiscale --
 *     filter:forward(frame) of both frames          getFilter(geometry), opticalflow_model.lua:45-79
 *     prepareInput(geometry, last_im, im)           :131-151: patch 1 narrowed to rows ceil(maxh/2) .. (H - maxh + 1 of them), columns alike
 *     model:forward(input)                          getModel(geometry, true, true), :81-129: SpatialMatching(maxh, maxw) -> Minus -> SoftMax over the window
 *     processOutput(geometry, moutput, true, thr)   :201-252: without threshold the first maximum with the centre tie-break (:153-161), confidences 1;
 *                                                   with one extractoutput.extractOutput(p, scores, 0.11, imaxs) and confidences = scores > thr (:163-167);
 *                                                   y, x = x2yx(index) - centered2onebased(0, 0); both pasted at floor((hImg - h) / 2), floor((wImg - w) / 2)
 *   I0, I1 [C][H][W] frames; layers: HOST array of the filter stack (C = layers[0].nIn).  nlayers == 0: I0 / I1 ARE the feature maps
 *   (geometry.prefilter: the caller keeps each frame's features for the next pair, as the script does) and patch 1's narrow is taken as a
 *   view of I0, no copy.  The output region is H1 x W1 = (H - hKernel + 1 - maxh + 1) x (W - wKernel + 1 - maxw + 1).
 *   Outputs (each may be NULL): full [2][hImg][wImg] (plane 0 = y, plane 1 = x; zero outside the pasted region), full_conf [hImg][wImg],
 *   index [H1][W1] int64 1-based class ids, scores [H1][W1] extractOutput's score (0 without threshold).  Pixels where no probability
 *   exceeds 0.11 -- imaxs / scores uninitialised in the reference -- get index = the centre class and score 0.
 *   16- / 17-wide windows on maps at least 253 columns wide never write the volume (the matcher's soft-max epilogue); other shapes go
 *   through the stand-alone ops.  Either way the results equal the module path's (getModel():forward + processOutput) bit for bit. */
int dfe_flow_pair_filtered_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, const dfe_filter_layer *layers, int nlayers,
                               int maxh, int maxw, int use_threshold, double threshold, int hImg, int wImg, float *full, float *full_conf,
                               int64_t *index, float *scores);
/* replaces: nn.SpatialMatching(maxh, maxw, false):forward({patch1, patch2}) where patch1 is the NARROW of a larger map
 *   (prepareInput, opticalflow_model.lua:147-149: `patch1:narrow(2, ..):narrow(3, ..)` is a view; nnx made it contiguous inside the
 *   module): in1 is read in place, rows in1_pitch floats apart and planes in1_plane floats apart.  Same output as dfe_spatial_matching_f32. */
int dfe_spatial_matching_strided_f32(dfe_ctx *ctx, const float *in1, int in1_pitch, int64_t in1_plane, const float *in2, int K, int H1, int W1,
                                     int maxh, int maxw, float *out);

/* ---- A4b: nn.CascadingAddTable:updateGradInput --------------------------------------------- */
/* replaces: CascadingAddTable.lua:137-154 (HEAD's graph has no trainable parameters in it: Mul2 / Power are
 *   commented out, :29,46,57 -- accGradParameters is a no-op).  gradOut[s], gradIn[s]: [P][maxh][maxw];
 *   g_0 = gradOut_0, g_{i+1} = gradOut_{i+1} + zero-pad(q x q block sums of g_i), gradIn_i = g_i. */
int dfe_cascading_add_backward_f32(dfe_ctx *ctx, const float *const *gradOut, const int *ratios, int nratios,
                                   int64_t P, int maxh, int maxw, float *const *gradIn);

/* ---- A2(upsample)+A4+A5: cascade and ring extraction for a whole frame --------------------- */
/* replaces: nearest upsampling of SpatialPyramid + CascadingAddTable + the "middle remover" +
 *   JoinTable(2) + SmartReshape(hImg,wImg,-2), opticalflow_model_multiscale.lua:227-229,285-333.
 *   prob[s] [H/r_s][W/r_s][maxh][maxw] (native scale).  out [H][W][nclasses]. */
int dfe_cascade_ring_f32(dfe_ctx *ctx, const float *const *prob, const int *ratios, int nratios,
                         int H, int W, int maxh, int maxw, float *out);

/* ---- A13 / A14: polar grids and bilinear warp ------------------------------------------------ */
/* replaces: getC2PMask radial/cartesian2polar.lua:4-49.  mask [2][hdst][wdst+lpadding+rpadding]. */
int dfe_polar_grid_c2p_f32(dfe_ctx *ctx, int wsrc, int hsrc, int wdst, int hdst, float xcenter,
                           float ycenter, int lpadding, int rpadding, float rmax, float alpha,
                           float *mask);
/* replaces: getP2CMask radial/cartesian2polar.lua:51-89.  mask [2][hdst][wdst]. */
int dfe_polar_grid_p2c_f32(dfe_ctx *ctx, int wsrc, int hsrc, int wdst, int hdst, float xcenter,
                           float ycenter, float rmax, float alpha, float *mask);
/* replaces: cartesian2polar(img, mask) = image.warp(img, mask, 'bilinear', false)
 *   radial/cartesian2polar.lua:91-93.  img [C][H][W], mask [2][Hd][Wd] (plane 0 = y, 1 = x, absolute,
 *   0-based), out [C][Hd][Wd]; coordinates outside the image are clamped. */
int dfe_warp_bilinear_f32(dfe_ctx *ctx, const float *img, int C, int H, int W, const float *mask,
                          int Hd, int Wd, float *out);

/* ---- A12(ii): radial flow -> depth --------------------------------------------------------- */
/* replaces: flow2depth radial/radial_opticalflow_display.lua:6-58.  rflow [H][W] scalar radial flow;
 *   depth = (d/f or infty)/infty where |p-c| > 10, conf = 0 inside that radius. */
int dfe_flow_to_depth_radial(dfe_ctx *ctx, const float *rflow, int H, int W, float xcenter,
                             float ycenter, float infty, float *depth, float *conf);

/* ---- A12(iii): x-flow -> depth of the drone API ----------------------------------------------- */
/* replaces: ARdroneAPI::computeDepthMapFromFlow ardrone/ardrone_api.cpp:99-140.  xflow, mask [H][W];
 *   mode filter of round(xflow) over the (sic) half-open 6x6 window [i-3,i+3) x [j-3,j+3) of pixels with non-zero
 *   mask (20 bins, values -8..11; samples outside are skipped -- the reference indexes its histogram unchecked),
 *   depth = imu_tx*|j-W/2|/|mode| (100 where |mode| < 1.1), conf = 1 where mask > 0.5 and j != W/2; elsewhere
 *   conf = 0 and depth = 0 (uninitialised in the reference). */
int dfe_flow_to_depth_ardrone(dfe_ctx *ctx, const float *xflow, const float *mask, int H, int W, float imu_tx,
                              float *depth, float *conf);

/* ---- A16: postProcessImage(input, mask, winsize, method) ------------------------------------ */
/* replaces: the inline-C `fmax` (mode) and `fmed` (median) filters and their Lua wrapper
 *   opticalflow_model.lua:323-472.  flow, out [2][H][W] (plane 0 = y, 1 = x), mask [H][W].
 *   method 0 = 'max': floor(flow+0.5), shift by the global minimum m, most frequent (vx,vy) of the masked
 *   k x k window (lowest code on ties), + m everywhere (the untouched border becomes m, as shipped :440);
 *   DFE_E_ARG when the rounded range exceeds the reference's 16 x 16 histogram.  method 1: per-component
 *   masked median tmp[n/2]; DFE_E_ARG when k*k > 32 (the reference's buffer).  Windows run over
 *   i < H-k, j < W-k as shipped.  Synchronises (method 0 reads the range back). */
int dfe_postprocess_image_f32(dfe_ctx *ctx, const float *flow, const float *mask, int H, int W,
                              int winsize, int method, float *out);

/* ---- A17: enlargeMask(mask, ix, iy), in place ------------------------------------------------- */
/* replaces: depth_estimation_api.lua:76-132. */
int dfe_enlarge_mask_f32(dfe_ctx *ctx, float *mask, int H, int W, int ix, int iy);

/* ---- A18: nn.OutputExtractor:updateOutput (soft arg-max) -------------------------------------- */
/* replaces: OutputExtractor.lua:21-35 (used by output_extraction_method = 'mean', opticalflow_model.lua:115-116).
 *   input [P][maxh*maxw] -> x [P] = sum p*j, y [P] = sum p*i with 1-based cell coordinates. */
int dfe_output_extractor_f32(dfe_ctx *ctx, const float *input, int64_t P, int maxh, int maxw,
                             float *x, float *y);

/* ---- A15 (next-row N1): the learned patch-feature stack -------------------------------------------- */
/* replaces: nn.SpatialConvolution / nn.SpatialConvolutionMap / nn.Tanh as used by getFilter
 *   (opticalflow_model.lua:45-79, radial/radial_opticalflow_network.lua:6-30): valid cross-correlation + bias,
 *   in [nIn][H][W] -> out [nOut][H-kH+1][W-kW+1]; weight [nOut][nIn][kH][kW] resp. one [kH][kW] kernel per
 *   connection of conn [nConn][2] = (from, to), 1-based, device int32; bias [nOut] or NULL.  Direct form, one thread
 *   per output, accumulation order (input plane | connection, ky, kx) -- correctness first, not tuned (this is where
 *   an implicit-GEMM MFMA kernel belongs). */
int dfe_spatial_convolution_f32(dfe_ctx *ctx, const float *in, const float *weight, const float *bias, int nIn, int nOut,
                                int H, int W, int kH, int kW, float *out);
/* replaces: nn.SpatialConvolution followed by nn.Tanh, as getFilter chains them (opticalflow_model.lua:48-64), in one launch; results equal
 *   dfe_spatial_convolution_f32 + dfe_tanh_f32 bit for bit. */
int dfe_spatial_convolution_tanh_f32(dfe_ctx *ctx, const float *in, const float *weight, const float *bias, int nIn, int nOut, int H, int W,
                                     int kH, int kW, float *out);
int dfe_spatial_convolution_map_f32(dfe_ctx *ctx, const float *in, const float *weight, const float *bias,
                                    const int32_t *conn, int nConn, int nIn, int nOut, int H, int W, int kH, int kW,
                                    float *out);
int dfe_tanh_f32(dfe_ctx *ctx, const float *in, int64_t n, float *out);
/* nn.SpatialContrastiveNormalization(nIn, kernel1d, threshold, thresval) (version2/network.lua:12, in front of both filter
 * branches): subtractive then divisive local normalisation with a separable 1-D kernel (k <= 33 taps, HOST pointer),
 * border-corrected by the estimator of a ones tensor.  Un-vendored nn, restated from recall: parity unpinned.
 * in / out [C][H][W]. */
int dfe_contrastive_normalization_f32(dfe_ctx *ctx, const float *in, int C, int H, int W, const float *kernel_host, int k,
                                      float threshold, float thresval, float *out);
/* ---- frames as the camera delivers them: uint8 planes ("upload frames as uint8, not fp32") ---------------------- */
/* dst[i] = float(src[i]) * scale (scale = 1: the integer values; 1/255: image.load's [0, 1] range). */
int dfe_u8_to_f32(dfe_ctx *ctx, const uint8_t *src, int64_t n, float scale, float *dst);
/* replaces: image.rgb2y as prepareInput calls it (opticalflow_model.lua:136-138; un-vendored `image`, restated: parity unpinned).
 *   rgb [3][H][W] -> y [1][H][W] = 0.299 R + 0.587 G + 0.114 B, accumulated in that order with separately rounded products and sums. */
int dfe_rgb2y_f32(dfe_ctx *ctx, const float *rgb, int H, int W, float *y);
/* replaces: tensor:min(1) on an n x M view (tests/time_matching.lua:41-43: `output:min(1)` of the matcher's output behind
 *   nn.Reshape(wsize*wsize, ...)): val[m] = min_r in[r][m], idx[m] = the first r attaining it, 1-based; either may be NULL. */
int dfe_min_dim0_f32(dfe_ctx *ctx, const float *in, int n, int64_t M, float *val, int64_t *idx);
/* dfe_flow_depth_pair_f32 on uint8 frames [C][H][W]: float(frame) * scale is made on the device (one conversion pass into a per-ctx
 * buffer) and the fp32 pipeline runs on it -- results are bit-identical to the fp32 entry on the same values.  A quarter of the
 * host-to-device bytes. */
int dfe_flow_depth_pair_u8(dfe_ctx *ctx, const uint8_t *I0, const uint8_t *I1, int C, int H, int W, int k, int hWin, int wWin,
                           float foe_x, float foe_y, double extract_threshold, float scale, float *flow, float *scores, float *depth,
                           float *depth_conf);
/* dfe_multiscale_flow_pair_f32 (f16_scale == 0) or dfe_multiscale_flow_pair_f16 (f16_scale > 0: its `scale`) on uint8 frames: no
 * conversion pass -- the pyramid's preparation kernels, the only readers of the frames, take the bytes as they are (float(byte) * scale
 * at the load), so the results are bit-identical to the fp32 entries on the converted frames. */
int dfe_multiscale_flow_pair_u8(dfe_ctx *ctx, const uint8_t *I0, const uint8_t *I1, int C, int H, int W, int k, int maxh, int maxw,
                                const int *ratios, int nratios, float scale, float f16_scale, float *flow, int64_t *idx);
/* ---- pipelined ingest: the frame loop's host boundary --------------------------------------------------------------------------- */
/* replaces: the serial load -> filter -> match of depth_estimation_opticalflow.lua:66-150 (loader:getNextFrame, then the model) where
 *   frames arrive in HOST memory.  dfe_ingest_submit_u8 starts the upload of a uint8 frame pair (nbytes per frame; pinned host memory --
 *   dfe_host_alloc / dfe_host_register -- for a copy-engine transfer that runs beside the kernels) into one of the ctx's three device
 *   slots on the ctx's own copy stream and returns the slot; dfe_flow_depth_pair_u8_slot runs dfe_flow_depth_pair_u8's step on that
 *   slot's frames once they have landed (the compute stream waits for the slot's event; no host synchronisation).  Submit pair i+1, then
 *   compute pair i: the next pair travels while the kernels of this one run.  A slot comes round again with every third submit; the
 *   submit then waits ON THE HOST until the compute stream has consumed the slot's previous pair (two pairs back: in a loop that stays
 *   one pair ahead it never actually waits).  Outputs as dfe_flow_depth_pair_u8 (bit-identical), ready in ctx-stream order. */
int dfe_ingest_submit_u8(dfe_ctx *ctx, const uint8_t *hI0, const uint8_t *hI1, int64_t nbytes, int *slot);
int dfe_flow_depth_pair_u8_slot(dfe_ctx *ctx, int slot, int C, int H, int W, int k, int hWin, int wWin, float foe_x, float foe_y,
                                double extract_threshold, float scale, float *flow, float *scores, float *depth, float *depth_conf);


/* ---- version2/: the single-scale learned model as ONE call -------------------------------------------------------------------- */
/* replaces: version2/test.lua:40-53 on getNetwork(datap) of version2/network.lua:5-39 for one frame pair --
 *   network:forward({prev, cur}) = ParallelTable{ Sequential{SpatialContrastiveNormalization(C, gaussian1D(normalization_k)),
 *   SpatialPadding(-lWin, -tWin, -rWin, -bWin), conv stack}, Sequential{the same normalisation, the same conv stack (shared weights)} }
 *   -> SpatialMatching(hWin, wWin, false); then `output:min(3)`, idx - 1, yflow = floor(idx / wWin) - tWin, xflow = idx - yflow' * wWin - lWin
 *   with lWin = ceil(wWin/2) - 1, tWin = ceil(hWin/2) - 1 (test.lua:18-21).  The minimum is taken over the whole hWin x wWin window in
 *   index order (first minimum), which is what the decode that follows it assumes.
 *   prev, cur [C][H][W]; norm_kernel_host: the 1-D normalisation kernel (HOST, norm_k <= 33 taps); layers: HOST array of nlayers
 *   convolution layers (version2 puts no Tanh between them: tanh_after = 0; the struct's other fields as in getFilter).
 *   Outputs over the dense inference region H1 x W1 = (H - (hWin-1) - (hKernel-1)) x (W - (wWin-1) - (wKernel-1)):
 *   xflow, yflow [H1][W1] float (may be NULL), idx [H1][W1] int64 1-based as torch's min returns it (may be NULL),
 *   volume [H1][W1][hWin][wWin] (may be NULL: then it lives in the ctx scratch arena only). */
int dfe_version2_flow_pair_f32(dfe_ctx *ctx, const float *prev, const float *cur, int C, int H, int W, const float *norm_kernel_host,
                               int norm_k, float threshold, float thresval, const dfe_filter_layer *layers, int nlayers, int hWin,
                               int wWin, float *xflow, float *yflow, int64_t *idx, float *volume);
/* replaces: nn.SpatialMatching(maxh, maxw, false) followed by `output:min(3)` and the index -> displacement decode, as
 *   version2/test.lua:45-51 and tests/time_matching.lua:18,41-43 run them on feature maps (in1 [K][H1][W1], in2 [K][H1+maxh-1][W1+maxw-1]):
 *   idx [H1][W1] int64 1-based first minimum of the window (index order), yflow = floor((idx-1) / maxw) - (ceil(maxh/2) - 1),
 *   xflow = (idx-1) mod maxw - (ceil(maxw/2) - 1); any of the three may be NULL.  The volume is never written where the flat-tile
 *   matcher takes the shape (16- / 17-wide windows on maps at least 253 columns wide); the results are those of
 *   dfe_spatial_matching_f32 + the first minimum, bit for bit. */
int dfe_spatial_matching_argmin_f32(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh, int maxw,
                                    int64_t *idx, float *xflow, float *yflow);
/* nn.SpatialConvolution [+ nn.Tanh] as an implicit GEMM on the matrix cores (v_mfma_f32_16x16x4_f32: f32 in, f32
 * accumulate, an fmaf chain in the reference's (input plane, ky, kx) order).  Same layouts as dfe_spatial_convolution_f32;
 * results differ from it by the fusing of each multiply-add only (<= 1e-5 relative to sum |terms|).  kH x kW up to what
 * fits one block's LDS (17 x 17 does); DFE_E_UNSUPPORTED beyond. */
int dfe_spatial_convolution_mfma_f32(dfe_ctx *ctx, const float *in, const float *weight, const float *bias, int nIn, int nOut,
                                     int H, int W, int kH, int kW, int tanh_after, float *out);

/* ---- next-row N2: gradients of the filter stack and of the soft-max / log layers, so that network:backward(input, df_do)
 *      reaches the convolution weights through the drop-in (radial/train_radial_opticalflow.lua:228-252,
 *      opticalflow.lua:296-338).  The modules are un-vendored nn; pinned by the finite-difference Jacobian of the
 *      forward (the method of tests/test_cascad.lua:21-25). ------------------------------------------------------ */
/* nn.SpatialConvolution:updateGradInput: gradIn[i][y][x] = sum_o sum_u sum_v w[o][i][u][v] gradOut[o][y-u][x-v];
 * gradOut [nOut][H-kH+1][W-kW+1], gradIn [nIn][H][W]. */
int dfe_spatial_convolution_grad_input_f32(dfe_ctx *ctx, const float *gradOut, const float *weight, int nIn, int nOut, int H,
                                           int W, int kH, int kW, float *gradIn);
/* nn.SpatialConvolution:accGradParameters(input, gradOutput, scale): gradWeight[o][i][u][v] += scale * sum_{y,x}
 * gradOut[o][y][x] in[i][y+u][x+v]; gradBias[o] += scale * sum gradOut[o] (gradBias may be NULL).  ACCUMULATES (zero the
 * buffers first, as zeroGradParameters does); shared between the two filter branches like the reference's clone(). */
int dfe_spatial_convolution_acc_grad_f32(dfe_ctx *ctx, const float *in, const float *gradOut, int nIn, int nOut, int H, int W,
                                         int kH, int kW, float scale, float *gradWeight, float *gradBias);
/* the same for nn.SpatialConvolutionMap (weight / gradWeight [nConn][kH][kW], conn [nConn][2] (from, to) 1-based, device int32) */
int dfe_spatial_convolution_map_grad_input_f32(dfe_ctx *ctx, const float *gradOut, const float *weight, const int32_t *conn,
                                               int nConn, int nIn, int nOut, int H, int W, int kH, int kW, float *gradIn);
int dfe_spatial_convolution_map_acc_grad_f32(dfe_ctx *ctx, const float *in, const float *gradOut, const int32_t *conn, int nConn,
                                             int nIn, int nOut, int H, int W, int kH, int kW, float scale, float *gradWeight,
                                             float *gradBias);
/* nn.Tanh:updateGradInput from the module's OUTPUT: gradIn = gradOut * (1 - out^2) */
int dfe_tanh_backward_f32(dfe_ctx *ctx, const float *out, const float *gradOut, int64_t n, float *gradIn);
/* nn.Log2 (Log.lua:13-28): forward clamps the INPUT in place to >= null_epsilon when clamp != 0 (:15-18), out = log(input);
 * backward gradIn = gradOut / input (the clamped one). */
int dfe_log2_forward_f32(dfe_ctx *ctx, float *input, int64_t n, float null_epsilon, int clamp, float *out);
int dfe_log2_backward_f32(dfe_ctx *ctx, const float *input, const float *gradOut, int64_t n, float *gradIn);
/* nn.LogSoftMax over rows [P][N] (radial/radial_opticalflow_network.lua:50) and its gradient gradIn = gradOut -
 * exp(out) * sum(gradOut); nn.SoftMax's gradient gradIn = out * (gradOut - sum(gradOut * out)) for the window soft-max
 * of getModel (opticalflow_model.lua:96-109; forward = dfe_softmin_f32 on the un-negated costs). */
int dfe_log_softmax_f32(dfe_ctx *ctx, const float *in, int64_t P, int N, float *out);
int dfe_log_softmax_backward_f32(dfe_ctx *ctx, const float *out, const float *gradOut, int64_t P, int N, float *gradIn);
int dfe_softmax_backward_f32(dfe_ctx *ctx, const float *out, const float *gradOut, int64_t P, int N, float *gradIn);

/* ---- A11 ('mean' extraction): marginal of the window over its columns --------------------------- */
/* replaces: input:reshape(H,W,maxh,maxw):sum(4) in getOutputConfidences2, opticalflow_model.lua:192.
 *   in [P][A][B] -> out [P][A], double accumulator as in TH. */
int dfe_marginal_sum_f32(dfe_ctx *ctx, const float *in, int64_t P, int A, int B, float *out);

/* ---- next-row N4: ego-motion rectification and the epipole / focus of expansion, the step in front of the polar warp
 *      (radial/radial_opticalflow_data.lua:211-231, depth_estimation_api.lua:139-147).  The reference calls the
 *      un-vendored, OpenCV-backed `sfm2`; restated from the calling convention -- parity unpinned. ---------------- */
/* e2 = K T / (K T)_3 * scale (data.lua:218-220: scale = networkp.wImg / calibrationp.wImg).  K row-major 3 x 3, T 3, host. */
int dfe_epipole(const double *K9, const double *T3, double scale, double *e2_xy);
/* sfm2.removeEgoMotion(img, K, R, 'bilinear') -> warped, mask: out(p) = bilinear(img, K R K^-1 p) (inverse != 0: R^T),
 * mask(p) = 1 where the source lies inside the frame, else 0 (and out = 0).  img / out [C][H][W], mask [H][W] or NULL;
 * K, R row-major host doubles. */
int dfe_remove_ego_motion_f32(dfe_ctx *ctx, const float *img, int C, int H, int W, const double *K9, const double *R9,
                              int inverse, float *out, float *mask);
/* sfm2.undistortImage(img, K, distP): inverse-map undistortion with the (k1, k2, p1, p2, k3) model of the .cal files. */
int dfe_undistort_image_f32(dfe_ctx *ctx, const float *img, int C, int H, int W, const double *K9, const double *dist5,
                            float *out);
/* NOT IN THE REFERENCE: an estimator of this library for the pure-translation case (the reference obtains the focus of expansion
 * as the epipole K T of sfm2.getEgoMotion2's pose: dfe_ego_motion_from_*_f32 + dfe_epipole below are that route).
 * Focus of expansion of a dense flow field (flow_y, flow_x [H][W]; conf [H][W] or NULL: pixels with conf <= 0 or
 * |flow| < min_flow are skipped): least-squares intersection of the flow lines, `iterations` Huber re-weightings (0..16).
 * foe_xy = (x, y) in pixels (host); n_used (may be NULL) = sum of the weights.  DFE_E_ARG when the lines are parallel. */
int dfe_foe_from_flow_f32(dfe_ctx *ctx, const float *flow_y, const float *flow_x, const float *conf, int H, int W,
                          float min_flow, int iterations, double *foe_xy, double *n_used);

/* ---- N4, the core: relative pose of the two frames from correspondences ------------------------------------------------------
 * replaces: sfm2.getEgoMotion2{im1, im2, K, maxPoints, pointsQuality, ransacMaxDist, pointsMinDistance} -> R, T, nFound, nInliers,
 *   fundmat (radial/radial_opticalflow_data.lua:211-217, radial/test_radial_opticalflow.lua:122-126; sfm2.getEgoMotion at
 *   depth_estimation_api.lua:141).  sfm2 (un-vendored, OpenCV) finds and tracks sparse corners itself; here the correspondences
 *   are an INPUT: pts1 / pts2 [N][2] = (x, y) pixel positions in the previous / current frame (device floats; weights [N] or NULL:
 *   entries <= 0 are skipped), or samples of the dense flow the matcher has produced (_from_flow: p2 = p1 + flow(p1) on a centred
 *   regular grid of at most max_points samples, conf <= 0 skipped).  `iterations` 8-point RANSAC hypotheses are built and scored
 *   in parallel on the device (Sampson distance <= ransac_max_dist pixels), the best is refitted over its inliers, projected onto
 *   the essential manifold and decomposed; cheirality picks among the four (R, T).  Restated from the calling convention: parity
 *   unpinned (3P).  Outputs (host doubles): R9 row-major, T3 (|T| = 1) with x2 ~ R x1 + T for camera coordinates of frame 1 in
 *   frame 2 -- so K T is the epipole in the current frame (data.lua:218) and dfe_remove_ego_motion_f32(prev, K, R, inverse = 1) undoes
 *   the rotation --, F9 = K^-T [T]x R K^-1 with unit Frobenius norm (may be NULL), n_found = usable samples, n_inliers.
 *   Deterministic for a given seed.  Synchronises.  DFE_E_ARG when no hypothesis reaches 8 inliers. */
int dfe_ego_motion_from_points_f32(dfe_ctx *ctx, const float *pts1, const float *pts2, const float *weights, int N, const double *K9,
                                   double ransac_max_dist, int iterations, unsigned seed, double *R9, double *T3, int *n_inliers,
                                   double *F9);
int dfe_ego_motion_from_flow_f32(dfe_ctx *ctx, const float *flow_y, const float *flow_x, const float *conf, int H, int W,
                                 const double *K9, int max_points, double ransac_max_dist, int iterations, unsigned seed, double *R9,
                                 double *T3, int *n_found, int *n_inliers, double *F9);

#ifdef __cplusplus
}
#endif
#endif
