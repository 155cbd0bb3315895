-- dfe_ffi.lua -- LuaJIT-FFI binding of libdfe.so (include/dfe.h) for the reference's Lua callers.
-- UNTESTED in the build image (no lua/luajit/th binary exists there); it is the binding a maintainer
-- adds on the reference side, line-for-line equivalent to the ctypes binding in ../_lib.py that the
-- test-suite exercises.  Needs LuaJIT-based Torch7 (tensor:data() returning cdata).
local ffi = require 'ffi'

ffi.cdef[[
typedef struct dfe_ctx dfe_ctx;
int dfe_ctx_create(int device, void *stream, int own_stream, dfe_ctx **out);
void dfe_ctx_destroy(dfe_ctx *ctx);
const char *dfe_last_error(const dfe_ctx *ctx);
int dfe_ctx_synchronize(dfe_ctx *ctx);
int dfe_malloc(dfe_ctx *ctx, size_t bytes, void **dptr);
int dfe_free(dfe_ctx *ctx, void *dptr);
int dfe_memcpy_h2d(dfe_ctx *ctx, void *dst, const void *src, size_t bytes);
int dfe_memcpy_d2h(dfe_ctx *ctx, void *dst, const void *src, size_t bytes);
int dfe_ssd_cost_volume_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W,
                            int kh, int kw, int hWin, int wWin, float *out);
int dfe_spatial_matching_f32(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1,
                             int W1, int maxh, int maxw, float *out);
int dfe_radial_matching_f32(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1,
                            int W, int hWin, float *out);
int dfe_argbest_center(dfe_ctx *ctx, const float *vol, int64_t P, int N, int middle, int take_max,
                       int64_t *idx, float *best);
int dfe_extract_output(dfe_ctx *ctx, const float *input, int H, int W, int N, float *scores,
                       double threshold, int64_t *imaxs);
int dfe_extract_output_marginalized(dfe_ctx *ctx, const float *input, int H, int W, int N,
                                    double threshold, double threshold_acc, int64_t *ret, int64_t *retgd);
int dfe_x2yx_multi(dfe_ctx *ctx, int maxh, int maxw, const int *ratios, int nratios,
                   const int64_t *idx, int64_t P, int64_t *y, int64_t *x, int compat_c);
int dfe_flow_depth_pair_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W,
                            int k, int hWin, int wWin, float foe_x, float foe_y,
                            double extract_threshold, float *flow, float *scores, float *depth,
                            float *depth_conf);
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

-- stage a (contiguous) host tensor on the device; returns a device pointer owned by the caller
function M.upload(t)
   t = t:contiguous()
   local bytes = t:nElement() * t:elementSize()
   local p = ffi.new('void*[1]')
   M.check(M.lib.dfe_malloc(M.ctx, bytes, p))
   M.check(M.lib.dfe_memcpy_h2d(M.ctx, p[0], t:data(), bytes))
   return p[0], bytes
end

function M.download(t, dptr)
   assert(t:isContiguous())
   M.check(M.lib.dfe_memcpy_d2h(M.ctx, t:data(), dptr, t:nElement() * t:elementSize()))
end

function M.free(dptr) M.check(M.lib.dfe_free(M.ctx, dptr)) end

return M
