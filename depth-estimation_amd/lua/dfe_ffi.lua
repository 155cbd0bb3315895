-- dfe_ffi.lua -- LuaJIT-FFI binding of libdfe.so (include/dfe.h) for the reference's Lua callers.
-- UNTESTED in the build image (no lua/luajit/th binary exists there); it is the binding a maintainer
-- adds on the reference side, line-for-line equivalent to the ctypes binding in ../_lib.py that the
-- test-suite exercises.  Needs LuaJIT-based Torch7 (tensor:data() returning cdata).
local ffi = require 'ffi'

ffi.cdef[[
typedef struct dfe_ctx dfe_ctx;
typedef struct dfe_radial_params { int C, hImg, wImg; int hInput, wInput; int hWin; int n1, kW1; int n2, kH2; int tanh_between; float alpha_polar; double kinfty; int zero_last_row; } dfe_radial_params;
typedef struct dfe_filter_layer { int nIn, nOut, kH, kW; const float *weight; const float *bias; const int32_t *conn; int nConn; int tanh_after; } dfe_filter_layer;
int dfe_version(void);
const char *dfe_kernel_revision(void);
int dfe_ctx_create(int device, void *stream, int own_stream, dfe_ctx **out);
void dfe_ctx_destroy(dfe_ctx *ctx);
const char *dfe_last_error(const dfe_ctx *ctx);
int dfe_ctx_synchronize(dfe_ctx *ctx);
void *dfe_ctx_stream(dfe_ctx *ctx);
int dfe_malloc(dfe_ctx *ctx, size_t bytes, void **dptr);
int dfe_free(dfe_ctx *ctx, void *dptr);
int dfe_memcpy_h2d(dfe_ctx *ctx, void *dst, const void *src, size_t bytes);
int dfe_memcpy_d2h(dfe_ctx *ctx, void *dst, const void *src, size_t bytes);
int dfe_host_register(dfe_ctx *ctx, void *ptr, size_t bytes);
int dfe_host_unregister(dfe_ctx *ctx, void *ptr);
int dfe_host_alloc(dfe_ctx *ctx, size_t bytes, void **hptr);
int dfe_host_free(dfe_ctx *ctx, void *hptr);
int dfe_set_cost_volume_kernel(dfe_ctx *ctx, int mode);
int dfe_set_cost_volume_tile(dfe_ctx *ctx, int tyq);
int dfe_set_option(dfe_ctx *ctx, const char *key, int value);
int dfe_get_option(dfe_ctx *ctx, const char *key, int *value);
const char *dfe_last_kernel(const dfe_ctx *ctx);
int dfe_set_scratch_limit(dfe_ctx *ctx, size_t bytes);
int dfe_device_alloc(dfe_ctx *ctx, size_t bytes, void **ptr, int *contiguous);
int dfe_device_free(dfe_ctx *ctx, void *ptr);
int dfe_profile_enable(dfe_ctx *ctx, int on);
int dfe_profile_read(dfe_ctx *ctx, double *total_ms, int *launches);
int dfe_profile_read_each(dfe_ctx *ctx, double *total_ms, int *launches, float *each_ms, int cap);
int dfe_stage_timers_enable(dfe_ctx *ctx, int on);
int dfe_stage_timers_read(dfe_ctx *ctx, double *ms , int *regions );
int dfe_ssd_cost_volume_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int kh, int kw, int hWin, int wWin, float *out);
int dfe_ssd_cost_volume_f16(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int kh, int kw, int hWin, int wWin, float scale, void *out);
int dfe_flow_depth_pair_f16(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int k, int hWin, int wWin, float foe_x, float foe_y, float scale, int64_t *idx, float *best, float *flow, float *depth, float *depth_conf);
int dfe_spatial_matching_f32(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh, int maxw, float *out);
int dfe_radial_matching_f32(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W, int hWin, float *out);
int dfe_radial_match_argmin_f32(dfe_ctx *ctx, const float *in1, int in1_plane_rows, const float *in2, int K, int H1, int W, int hWin, float *volume, float *flow, int zero_last_row);
int dfe_radial_out_shape(const dfe_radial_params *p, int *hMatch, int *hOut, int *wOut);
int dfe_radial_flow_depth_pair_f32(dfe_ctx *ctx, const dfe_radial_params *p, const float *prev, const float *cur, double e2x, double e2y, const float *w1, const float *b1, const float *w2, const float *b2, float *volume, float *polar_flow, float *cart_flow, float *depth, float *conf);
int dfe_spatial_matching_backward_f32(dfe_ctx *ctx, const float *in1, const float *in2, const float *gradOut, int K, int H1, int W1, int maxh, int maxw, float *gradIn1, float *gradIn2);
int dfe_radial_matching_backward_f32(dfe_ctx *ctx, const float *in1, const float *in2, const float *gradOut, int K, int H1, int W, int hWin, float *gradIn1, float *gradIn2);
int dfe_argbest_center(dfe_ctx *ctx, const float *vol, int64_t P, int N, int middle, int take_max, int64_t *idx, float *best);
int dfe_extract_output(dfe_ctx *ctx, const float *input, int H, int W, int N, float *scores, double threshold, int64_t *imaxs);
int dfe_extract_output_marginalized(dfe_ctx *ctx, const float *input, int H, int W, int N, double threshold, double threshold_acc, int64_t *ret, int64_t *retgd);
int dfe_x2yx(dfe_ctx *ctx, const int64_t *idx, int64_t P, int maxh, int maxw, int64_t *y, int64_t *x);
int dfe_x2yx_multi(dfe_ctx *ctx, int maxh, int maxw, const int *ratios, int nratios, const int64_t *idx, int64_t P, int64_t *y, int64_t *x, int compat_c);
int64_t dfe_yx2x_multi(int maxh, int maxw, const int *ratios, int nratios, double y, double x);
int dfe_x2yx_multi_number(int maxh, int maxw, const int *ratios, int nratios, int64_t id, int64_t *y, int64_t *x);
int64_t dfe_multi_nclasses(int maxh, int maxw, const int *ratios, int nratios);
int dfe_ssd_flow_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int kh, int kw, int hWin, int wWin, double extract_threshold, int64_t *idx, float *best, float *flow_y, float *flow_x, float *scores, int64_t *imaxs);
int dfe_flow_tail(dfe_ctx *ctx, const float *vol, int rows, int Wo, int hWin, int wWin, double threshold, int row_off, int64_t *idx, float *best, float *fy, float *fx, float *scores, int64_t *imaxs, int pitch, int pad_t, int pad_l, int scores_padded);
int dfe_flow_to_depth_cartesian(dfe_ctx *ctx, const float *flow, int H, int W, float cx, float cy, int fix_dot, float *depth, float *conf);
int dfe_flow_depth_pair_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int k, int hWin, int wWin, float foe_x, float foe_y, double extract_threshold, float *flow, float *scores, float *depth, float *depth_conf);
int dfe_downsample_box_f32(dfe_ctx *ctx, const float *img, int C, int H, int W, int r, float *out);
int dfe_pyramid_scale_volume_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int r, int kh, int kw, int maxh, int maxw, float *out);
int dfe_softmin_f32(dfe_ctx *ctx, const float *cost, int64_t P, int N, float *prob);
int dfe_cascading_add_f32(dfe_ctx *ctx, const float *const *in, const int *ratios, int nratios, int64_t P, int maxh, int maxw, float *const *out);
int dfe_cascade_flow_f32(dfe_ctx *ctx, const float *const *prob, const int *ratios, int nratios, int H, int W, int maxh, int maxw, int64_t *idx, float *best, float *flow_y, float *flow_x);
int dfe_multiscale_flow_pair_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int k, int maxh, int maxw, const int *ratios, int nratios, float *flow, int64_t *idx);
int dfe_multiscale_flow_pair_f16(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int k, int maxh, int maxw, const int *ratios, int nratios, float scale, float *flow, int64_t *idx);
int dfe_multiscale_flow_pair_filtered_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int maxh, int maxw, const int *ratios, int nratios, const dfe_filter_layer *layers, int nlayers, int share_filters, float f16_scale, float *flow, int64_t *idx);
int dfe_flow_pair_filtered_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, const dfe_filter_layer *layers, int nlayers, int maxh, int maxw, int use_threshold, double threshold, int hImg, int wImg, float *full, float *full_conf, int64_t *index, float *scores);
int dfe_spatial_matching_strided_f32(dfe_ctx *ctx, const float *in1, int in1_pitch, int64_t in1_plane, const float *in2, int K, int H1, int W1, int maxh, int maxw, float *out);
int dfe_cascading_add_backward_f32(dfe_ctx *ctx, const float *const *gradOut, const int *ratios, int nratios, int64_t P, int maxh, int maxw, float *const *gradIn);
int dfe_cascade_ring_f32(dfe_ctx *ctx, const float *const *prob, const int *ratios, int nratios, int H, int W, int maxh, int maxw, float *out);
int dfe_polar_grid_c2p_f32(dfe_ctx *ctx, int wsrc, int hsrc, int wdst, int hdst, float xcenter, float ycenter, int lpadding, int rpadding, float rmax, float alpha, float *mask);
int dfe_polar_grid_p2c_f32(dfe_ctx *ctx, int wsrc, int hsrc, int wdst, int hdst, float xcenter, float ycenter, float rmax, float alpha, float *mask);
int dfe_warp_bilinear_f32(dfe_ctx *ctx, const float *img, int C, int H, int W, const float *mask, int Hd, int Wd, float *out);
int dfe_flow_to_depth_radial(dfe_ctx *ctx, const float *rflow, int H, int W, float xcenter, float ycenter, float infty, float *depth, float *conf);
int dfe_flow_to_depth_ardrone(dfe_ctx *ctx, const float *xflow, const float *mask, int H, int W, float imu_tx, float *depth, float *conf);
int dfe_postprocess_image_f32(dfe_ctx *ctx, const float *flow, const float *mask, int H, int W, int winsize, int method, float *out);
int dfe_enlarge_mask_f32(dfe_ctx *ctx, float *mask, int H, int W, int ix, int iy);
int dfe_output_extractor_f32(dfe_ctx *ctx, const float *input, int64_t P, int maxh, int maxw, float *x, float *y);
int dfe_spatial_convolution_f32(dfe_ctx *ctx, const float *in, const float *weight, const float *bias, int nIn, int nOut, int H, int W, int kH, int kW, float *out);
int dfe_spatial_convolution_tanh_f32(dfe_ctx *ctx, const float *in, const float *weight, const float *bias, int nIn, int nOut, int H, int W, int kH, int kW, float *out);
int dfe_spatial_convolution_map_f32(dfe_ctx *ctx, const float *in, const float *weight, const float *bias, const int32_t *conn, int nConn, int nIn, int nOut, int H, int W, int kH, int kW, float *out);
int dfe_tanh_f32(dfe_ctx *ctx, const float *in, int64_t n, float *out);
int dfe_contrastive_normalization_f32(dfe_ctx *ctx, const float *in, int C, int H, int W, const float *kernel_host, int k, float threshold, float thresval, float *out);
int dfe_u8_to_f32(dfe_ctx *ctx, const uint8_t *src, int64_t n, float scale, float *dst);
int dfe_rgb2y_f32(dfe_ctx *ctx, const float *rgb, int H, int W, float *y);
int dfe_min_dim0_f32(dfe_ctx *ctx, const float *in, int n, int64_t M, float *val, int64_t *idx);
int dfe_flow_depth_pair_u8(dfe_ctx *ctx, const uint8_t *I0, const uint8_t *I1, int C, int H, int W, int k, int hWin, int wWin, float foe_x, float foe_y, double extract_threshold, float scale, float *flow, float *scores, float *depth, float *depth_conf);
int dfe_multiscale_flow_pair_u8(dfe_ctx *ctx, const uint8_t *I0, const uint8_t *I1, int C, int H, int W, int k, int maxh, int maxw, const int *ratios, int nratios, float scale, float f16_scale, float *flow, int64_t *idx);
int dfe_ingest_submit_u8(dfe_ctx *ctx, const uint8_t *hI0, const uint8_t *hI1, int64_t nbytes, int *slot);
int dfe_flow_depth_pair_u8_slot(dfe_ctx *ctx, int slot, int C, int H, int W, int k, int hWin, int wWin, float foe_x, float foe_y, double extract_threshold, float scale, float *flow, float *scores, float *depth, float *depth_conf);
int dfe_version2_flow_pair_f32(dfe_ctx *ctx, const float *prev, const float *cur, int C, int H, int W, const float *norm_kernel_host, int norm_k, float threshold, float thresval, const dfe_filter_layer *layers, int nlayers, int hWin, int wWin, float *xflow, float *yflow, int64_t *idx, float *volume);
int dfe_spatial_matching_argmin_f32(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh, int maxw, int64_t *idx, float *xflow, float *yflow);
int dfe_spatial_convolution_mfma_f32(dfe_ctx *ctx, const float *in, const float *weight, const float *bias, int nIn, int nOut, int H, int W, int kH, int kW, int tanh_after, float *out);
int dfe_spatial_convolution_grad_input_f32(dfe_ctx *ctx, const float *gradOut, const float *weight, int nIn, int nOut, int H, int W, int kH, int kW, float *gradIn);
int dfe_spatial_convolution_acc_grad_f32(dfe_ctx *ctx, const float *in, const float *gradOut, int nIn, int nOut, int H, int W, int kH, int kW, float scale, float *gradWeight, float *gradBias);
int dfe_spatial_convolution_map_grad_input_f32(dfe_ctx *ctx, const float *gradOut, const float *weight, const int32_t *conn, int nConn, int nIn, int nOut, int H, int W, int kH, int kW, float *gradIn);
int dfe_spatial_convolution_map_acc_grad_f32(dfe_ctx *ctx, const float *in, const float *gradOut, const int32_t *conn, int nConn, int nIn, int nOut, int H, int W, int kH, int kW, float scale, float *gradWeight, float *gradBias);
int dfe_tanh_backward_f32(dfe_ctx *ctx, const float *out, const float *gradOut, int64_t n, float *gradIn);
int dfe_log2_forward_f32(dfe_ctx *ctx, float *input, int64_t n, float null_epsilon, int clamp, float *out);
int dfe_log2_backward_f32(dfe_ctx *ctx, const float *input, const float *gradOut, int64_t n, float *gradIn);
int dfe_log_softmax_f32(dfe_ctx *ctx, const float *in, int64_t P, int N, float *out);
int dfe_log_softmax_backward_f32(dfe_ctx *ctx, const float *out, const float *gradOut, int64_t P, int N, float *gradIn);
int dfe_softmax_backward_f32(dfe_ctx *ctx, const float *out, const float *gradOut, int64_t P, int N, float *gradIn);
int dfe_marginal_sum_f32(dfe_ctx *ctx, const float *in, int64_t P, int A, int B, float *out);
int dfe_epipole(const double *K9, const double *T3, double scale, double *e2_xy);
int dfe_remove_ego_motion_f32(dfe_ctx *ctx, const float *img, int C, int H, int W, const double *K9, const double *R9, int inverse, float *out, float *mask);
int dfe_undistort_image_f32(dfe_ctx *ctx, const float *img, int C, int H, int W, const double *K9, const double *dist5, float *out);
int dfe_foe_from_flow_f32(dfe_ctx *ctx, const float *flow_y, const float *flow_x, const float *conf, int H, int W, float min_flow, int iterations, double *foe_xy, double *n_used);
int dfe_ego_motion_from_points_f32(dfe_ctx *ctx, const float *pts1, const float *pts2, const float *weights, int N, const double *K9, double ransac_max_dist, int iterations, unsigned seed, double *R9, double *T3, int *n_inliers, double *F9);
int dfe_ego_motion_from_flow_f32(dfe_ctx *ctx, const float *flow_y, const float *flow_x, const float *conf, int H, int W, const double *K9, int max_points, double ransac_max_dist, int iterations, unsigned seed, double *R9, double *T3, int *n_found, int *n_inliers, double *F9);
]]

local M = {}
M.lib = ffi.load('dfe')

local ctxp = ffi.new('dfe_ctx*[1]')
local rc = M.lib.dfe_ctx_create(0, nil, 1, ctxp)
if rc ~= 0 then error('libdfe: ' .. ffi.string(M.lib.dfe_last_error(nil))) end
M.ctx = ctxp[0]

-- non-zero return -> Lua error, the way the reference's modules fail (CascadingAddTable.lua:111,115,123)
function M.check(rc)
   if rc ~= 0 then error('libdfe: ' .. ffi.string(M.lib.dfe_last_error(M.ctx))) end
end

-- Device buffers are PERSISTENT: one per (module instance, role), grown when the tensor it mirrors grows, never freed per
-- call (a VGA SpatialMatching would otherwise spend far longer in hipMalloc / hipFree than in its kernel).
local Buffer = {}
Buffer.__index = Buffer
function M.newBuffer() return setmetatable({ptr = nil, bytes = 0}, Buffer) end
function Buffer:reserve(bytes)
   if bytes > self.bytes then
      if self.ptr ~= nil then M.check(M.lib.dfe_free(M.ctx, self.ptr)) end
      local p = ffi.new('void*[1]')
      M.check(M.lib.dfe_malloc(M.ctx, bytes, p))
      self.ptr, self.bytes = p[0], bytes
   end
   return self.ptr
end
function Buffer:free()
   if self.ptr ~= nil then M.check(M.lib.dfe_free(M.ctx, self.ptr)) end
   self.ptr, self.bytes = nil, 0
end

-- Host <-> device copies go through ONE library-owned pinned bounce buffer (dfe_host_alloc = hipHostMalloc), grown on demand
-- and reused by every module: direct DMA without ever pinning memory this binding does not own.  (The first version pinned
-- the tensors' own storage in place through dfe_host_register -- including the temporaries of :contiguous() / :long(), which
-- the garbage collector frees while they are still registered; a later tensor at the same address then inherited a stale
-- registration that maps the OLD physical pages.  Torch tensor storage is never registered any more.)  dfe_memcpy_* are
-- synchronous, so one bounce buffer serves all copies of the single Lua thread.
local bounce = {ptr = nil, bytes = 0}
local function bounce_reserve(bytes)
   if bytes > bounce.bytes then
      if bounce.ptr ~= nil then M.check(M.lib.dfe_host_free(M.ctx, bounce.ptr)) end
      bounce.ptr, bounce.bytes = nil, 0
      local want = math.max(bytes, 2 * bounce.bytes, 1048576)
      local p = ffi.new('void*[1]')
      M.check(M.lib.dfe_host_alloc(M.ctx, want, p))
      bounce.ptr, bounce.bytes = p[0], want
   end
   return bounce.ptr
end

-- the C ABI is typed: a DoubleTensor uploaded byte for byte would be read as floats
function M.checktype(t, typename, what)
   if torch.typename(t) ~= typename then
      error(string.format('%s must be a %s, got %s', what, typename, torch.typename(t) or type(t)))
   end
end

-- host tensor -> persistent device buffer `buf`; returns the device pointer
function M.upload(t, buf)
   t = t:contiguous()
   local bytes = t:nElement() * t:elementSize()
   local d = buf:reserve(bytes)
   if bytes > 0 then
      local h = bounce_reserve(bytes)
      ffi.copy(h, t:data(), bytes)
      M.check(M.lib.dfe_memcpy_h2d(M.ctx, d, h, bytes))
   end
   return d, bytes
end

function M.download(t, dptr)
   assert(t:isContiguous())
   local bytes = t:nElement() * t:elementSize()
   if bytes > 0 then
      local h = bounce_reserve(bytes)
      M.check(M.lib.dfe_memcpy_d2h(M.ctx, h, dptr, bytes))
      ffi.copy(t:data(), h, bytes)
   end
end

return M
