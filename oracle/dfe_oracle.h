/*
 * dfe_oracle.h -- CPU restatement of the reference's dense patch-correlation
 * flow->depth hot path (MichaelMathieu/depth-estimation).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under depth-estimation_amd/ (the product)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / the baseline.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - The reference is Lua on Torch7; no Lua/Torch7 runtime or headers exist
 *     in the build image, and its two native files need luaT.h / TH.h / lua.h
 *     (absent).  Nothing of the reference can be compiled or run here without
 *     writing stand-in headers, so oracle/_ref does not exist.
 *   - The restatement is pinned by the reference's own known-answer tests that
 *     need no data files: cartesian_groundtruth_cc_testme
 *     (radial/radial_opticalflow_groundtruth.lua:170-210), the codec round trip
 *     and brute-force SSD cross-check of tests/test_multiscale.lua:57-80,
 *     135-166, and a hand-derived extractOutput vector.  Pieces whose
 *     arithmetic lives in un-vendored nnx/nn (SoftMax numerics, down-sampling
 *     edge policy, image.warp border policy) are "parity unpinned".
 *
 * All tensors are dense row-major; sizes in elements.  Class ids are 1-based
 * exactly as in the reference.  "ref:" comments give reference file:line.
 */
#ifndef DFE_ORACLE_H
#define DFE_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_RATIOS 10 /* ref: x2yxMulti2.c:1 N_MAX_RATIOS */

void orc_set_num_threads(int n); /* ref: openmp.setDefaultNumThreads, depth_estimation_opticalflow.lua:40 */
int orc_get_max_threads(void);

/* A0 unfold (im2col). ref: radial/radial_opticalflow_groundtruth.lua:9-21
 * img [C][H][W] -> out [C*kh*kw][H-kh+1][W-kw+1], feature (c,i,j) = c*kh*kw+i*kw+j */
void orc_unfold(const float *img, int C, int H, int W, int kh, int kw, float *out);

/* A1 nn.SpatialMatching(maxh,maxw,false) on feature maps.
 * ref call sites: opticalflow_model.lua:93, radial/radial_opticalflow_groundtruth.lua:83;
 * semantics pinned by tests/test_multiscale.lua:149-166.
 * in1 [K][H1][W1], in2 [K][H1+maxh-1][W1+maxw-1] -> out [H1][W1][maxh][maxw] */
void orc_spatial_matching(const float *in1, const float *in2, int K, int H1, int W1,
                          int maxh, int maxw, float *out);

/* A15 / N1 learned filter stack: nn.SpatialConvolution, nn.SpatialConvolutionMap (connection table [nConn][2] =
 * (from, to), 1-based), nn.Tanh. ref: opticalflow_model.lua:45-79, radial/radial_opticalflow_network.lua:6-30.
 * Summation order of un-vendored nn recalled, not pinned by a reference test. */
void orc_spatial_convolution(const float *in, const float *weight, const float *bias, int nIn, int nOut, int H, int W,
                             int kH, int kW, float *out);
void orc_spatial_convolution_map(const float *in, const float *weight, const float *bias, const int *conn, int nConn,
                                 int nIn, int nOut, int H, int W, int kH, int kW, float *out);
void orc_spatial_convolution_fma(const float *in, const float *weight, const float *bias, int nIn, int nOut, int H, int W,
                                 int kH, int kW, float *out);
void orc_contrastive_normalization(const float *in, int C, int H, int W, const float *kernel, int k, float threshold,
                                   float thresval, float *out);
void orc_tanh(const float *in, int64_t n, float *out);
void orc_rgb2y(const float *rgb, int H, int W, float *y);
void orc_spatial_convolution_grad_input(const float *go, const float *weight, const int *conn, int nConn, int nIn, int nOut,
                                        int H, int W, int kH, int kW, float *gi);
void orc_spatial_convolution_acc_grad(const float *in, const float *go, const int *conn, int nConn, int nIn, int nOut, int H,
                                      int W, int kH, int kW, float scale, float *gw, float *gb);
void orc_tanh_backward(const float *out, const float *go, int64_t n, float *gi);
void orc_log_softmax(const float *in, int64_t P, int N, float *out);
void orc_log_softmax_backward(const float *out, const float *go, int64_t P, int N, float *gi);
void orc_softmax_backward(const float *out, const float *go, int64_t P, int N, float *gi);

/* N2: gradients of A1 and A1r w.r.t. in1 and in2 (go = gradOutput, layout of the forward output). Pinned as the Jacobian
 * of the forward restatements (no reference test exists; method of tests/test_cascad.lua:22). */
void orc_spatial_matching_backward(const float *in1, const float *in2, const float *go, int K, int H1, int W1,
                                   int maxh, int maxw, float *g1, float *g2);
void orc_radial_matching_backward(const float *in1, const float *in2, const float *go, int K, int H1, int W,
                                  int hWin, float *g1, float *g2);

/* A0+A1 fused on raw frames (never materialises the unfolded features):
 * unfold(kh,kw) -> crop frame0 features by floor/ceil((win-1)/2) -> SpatialMatching.
 * ref: radial/radial_opticalflow_groundtruth.lua:79-84.
 * rows [row0,row1) of the Ho x Wo output are computed (row0=0,row1=Ho for all);
 * out is the full [Ho][Wo][hWin][wWin] buffer. Ho=H-kh+1-hWin+1, Wo=W-kw+1-wWin+1 */
void orc_ssd_cost_volume(const float *I0, const float *I1, int C, int H, int W,
                         int kh, int kw, int hWin, int wWin, float *out,
                         int row0, int row1);

/* A1r nn.SpatialRadialMatching(hWin). ref: radial/radial_opticalflow_network.lua:33,59-72
 * in1 [K][H1][W], in2 [K][H1+hWin-1][W] -> out [H1][W][hWin] */
void orc_radial_matching(const float *in1, const float *in2, int K, int H1, int W,
                         int hWin, float *out);

/* A6 arg-min / arg-max over the last dim with the centre tie-break.
 * ref: radial/radial_opticalflow_groundtruth.lua:88-94 (min), opticalflow_model.lua:153-161 (max).
 * vol [P][N]; idx 1-based; first index wins ties; if best == vol[p][middle-1] idx=middle.
 * middle<=0 disables the tie-break. */
void orc_argbest_center(const float *vol, int64_t P, int N, int middle, int take_max,
                        int64_t *idx, float *best);

/* A7 extractoutput.extractOutput. ref: version2/extract_output.cpp:63-155 (root copy :63-155)
 * input [H*W][N]; imaxs/scores are left untouched for pixels with nothing > threshold. */
void orc_extract_output(const float *input, int64_t P, int N, double threshold,
                        int64_t *imaxs, float *scores);
/* A8 extractOutputMarginalized. ref: version2/extract_output.cpp:157-255 */
void orc_extract_output_marginalized(const float *input, int64_t P, int N, double threshold,
                                     double threshold_acc, int64_t *ret, int64_t *retgd);

/* A9 single-scale decode. ref: radial/radial_opticalflow_groundtruth.lua:97-100,
 * opticalflow_model.lua:16-34,209-213 */
void orc_x2yx(const int64_t *idx, int64_t P, int maxh, int maxw, int64_t *y, int64_t *x);

/* A10 multiscale codec, Lua scalar semantics.
 * ref: opticalflow_model_multiscale.lua:10-52 (yx2xMulti), :83-132 (x2yxMultiNumber) */
int64_t orc_yx2x_multi(int maxh, int maxw, const int *ratios, int nratios, double y, double x);
int orc_x2yx_multi_number(int maxh, int maxw, const int *ratios, int nratios, int64_t id,
                          int64_t *y, int64_t *x);
int orc_x2yx_multi(int maxh, int maxw, const int *ratios, int nratios, const int64_t *idx,
                   int64_t P, int64_t *y, int64_t *x);
int64_t orc_multi_nclasses(int maxh, int maxw, const int *ratios, int nratios);
/* bug-compatible restatement of the shipped vectorised C body. ref: x2yxMulti2.c:1-95.
 * ids it never matches leave y/x untouched (as shipped). */
void orc_x2yx_multi_compat_c(int maxh, int maxw, const int *ratios, int nratios,
                             const int64_t *idx, int64_t P, int64_t *y, int64_t *x);

/* A2 pieces. ref: opticalflow_model_multiscale.lua:134-173 (getMultiscalePrefilter), :196-229 */
void orc_downsample_box(const float *img, int C, int H, int W, int r, float *out); /* [C][H/r][W/r] */
void orc_zero_pad(const float *img, int C, int H, int W, int pl, int pr, int pt, int pb, float *out);
/* one scale of the pyramid on raw frames: downsample by r, zero-pad by hPatch2-1,
 * crop frame0 by maxw-1, kxk raw-patch features, SpatialMatching(maxh,maxw).
 * out [H/r][W/r][maxh][maxw] (native scale, not upsampled) */
void orc_pyramid_scale_volume(const float *I0, const float *I1, int C, int H, int W, int r,
                              int kh, int kw, int maxh, int maxw, float *out);

/* A3 per-pixel softmin over N cells: p = softmax(-cost). ref: opticalflow_model_multiscale.lua:270-279 */
void orc_softmin(const float *cost, int64_t P, int N, float *prob);

/* A2(upsample)+A4+A5: cascade + ring extraction for one finest-scale image.
 * prob[s] is [H/r_s][W/r_s][maxh][maxw] (native scale); nearest-neighbour upsample to HxW is
 * implied (pixel (y,x) reads (y/r, x/r)). ref: CascadingAddTable.lua:108-135,
 * opticalflow_model_multiscale.lua:293-333.  out [H][W][nclasses] */
int orc_cascade_ring(const float *const *prob, int nratios, const int *ratios, int H, int W,
                     int maxh, int maxw, float *out);
/* A4 alone on already-upsampled windows: in[s] [P][maxh][maxw] -> out[s] same. ref: CascadingAddTable.lua:108-135 */
int orc_cascading_add(const float *const *in, int nratios, const int *ratios, int64_t P,
                      int maxh, int maxw, float *const *out);

/* A4b gradient of A4 w.r.t. its inputs: gradOut[s], gradIn[s] [P][maxh][maxw]. ref: CascadingAddTable.lua:137-154;
 * pinned as the adjoint of orc_cascading_add (the reference's own test is a Jacobian check, tests/test_cascad.lua:22) */
int orc_cascading_add_backward(const float *const *gradOut, int nratios, const int *ratios, int64_t P,
                               int maxh, int maxw, float *const *gradIn);

/* A11 centre-paste. ref: opticalflow_model.lua:227-250 */
void orc_paste_center(const float *src, int h, int w, float *dst, int H, int W);

/* A12 flow -> depth.
 * (i) ref: test_opticalflow.lua:143-216 `radial` (dot-product quirk :181 replicated unless fix_dot)
 * flow [2][H][W] (plane0=y, plane1=x), centre (cx,cy) */
void orc_flow_to_depth_cartesian(const float *flow, int H, int W, float cx, float cy,
                                 int fix_dot, float *depth, float *conf);
/* (ii) ref: radial/radial_opticalflow_display.lua:6-58 `flow2depth` */
void orc_flow_to_depth_radial(const float *rflow, const float *cartidx_conf, int H, int W,
                              float cx, float cy, float infty, float *depth, float *conf);

/* A11 'mean' extraction: marginal of the window over its columns. ref: opticalflow_model.lua:192. in [P][A][B] -> [P][A] */
void orc_marginal_sum(const float *in, int64_t P, int A, int B, float *out);

/* (iii) ref: ardrone/ardrone_api.cpp:99-140 `computeDepthMapFromFlow`: xflow, mask [H][W]; m = IMU x-translation */
void orc_flow_to_depth_ardrone(const float *xflow, const float *mask, int H, int W, float m, float *depth, float *conf);

/* A13 polar grids. ref: radial/cartesian2polar.lua:4-49 (C2P), :51-89 (P2C) */
void orc_polar_grid_c2p(int wsrc, int hsrc, int wdst, int hdst, float xc, float yc,
                        int lpad, int rpad, float rmax, float alpha, float *mask /*[2][hdst][wdst+lpad+rpad]*/);
void orc_polar_grid_p2c(int wsrc, int hsrc, int wdst, int hdst, float xc, float yc,
                        float rmax, float alpha, float *mask /*[2][hdst][wdst]*/);
/* A14 absolute-coordinate bilinear warp (image.warp(img,mask,'bilinear',false)).
 * ref: radial/cartesian2polar.lua:91-93.  Border policy is un-vendored: this restatement
 * clamps coordinates to the image (parity unpinned). */
void orc_warp_bilinear(const float *img, int C, int H, int W, const float *mask, int Hd, int Wd,
                       float *out);

/* A16 postProcessImage(input, mask, winsize, method). ref: opticalflow_model.lua:323-472.
 * flow [2][H][W] (plane0=y, plane1=x), mask [H][W], out [2][H][W].
 * method 0 = 'max': mode filter of the rounded flow (fmax :342-386) incl. the +0.5 floor, global-min shift and
 * the final +m that also lifts the untouched border (:436-440); returns -1 if the rounded range exceeds the
 * reference's 16x16 histogram.  method 1: per-component masked median (fmed :388-434); -1 if k*k > 32. */
int orc_postprocess_image(const float *flow, const float *mask, int H, int W, int k, int method, float *out);
/* A17 enlargeMask(mask, ix, iy), in place. ref: depth_estimation_api.lua:76-132 */
void orc_enlarge_mask(float *mask, int H, int W, int ix, int iy);
/* A18 nn.OutputExtractor:updateOutput (soft arg-max expectation). ref: OutputExtractor.lua:21-35.
 * input [P][maxh*maxw] -> x[P] = sum_k input*j(k), y[P] = sum_k input*i(k), 1-based cell coordinates */
void orc_output_extractor(const float *input, int64_t P, int maxh, int maxw, float *x, float *y);

/* ---- next-row N4 (call sites only: `sfm2` is un-vendored; parity unpinned) ------------------------------------------ */
int orc_epipole(const double *K9, const double *T3, double scale, double *e2_xy);   /* ref: radial/radial_opticalflow_data.lua:218-220 */
int orc_remove_ego_motion(const float *img, int C, int H, int W, const double *K9, const double *R9, int inverse, float *out,
                          float *mask); /* ref: radial/radial_opticalflow_data.lua:231, depth_estimation_api.lua:147 */
void orc_undistort_image(const float *img, int C, int H, int W, const double *K9, const double *dist5,
                         float *out);   /* ref: radial/radial_opticalflow_data.lua:24, depth_estimation_api.lua:139 */
int orc_foe_from_flow(const float *fy, const float *fx, const float *conf, int H, int W, float min_flow, int iterations, double *foe_xy,
                      double *n_used);  /* not in the reference: counterpart of dfe_foe_from_flow_f32 */
int orc_ego_motion_from_points(const float *pts1, const float *pts2, const float *weights, int N, const double *K9, double ransac_max_dist,
                               int iterations, unsigned seed, double *R9, double *T3, int *n_inliers,
                               double *F9); /* ref: radial/radial_opticalflow_data.lua:211-217, depth_estimation_api.lua:141 */

#ifdef __cplusplus
}
#endif
#endif
