-- Drop-in for the global x2yxMulti2(geometry, x) of opticalflow_model_multiscale.lua:72-81 (which compiles x2yxMulti2.c
-- through `inline` at every call): class ids -> (y, x) displacements on the device.  UNTESTED here (no Lua runtime in the
-- build image).
-- Default (compat_c = 0, the same in the Python mirror and in the fused cascade kernels): the Lua scalar codec
-- x2yxMultiNumber that the reference's own round-trip test pins (tests/test_multiscale.lua:57-80).  compat_c = 1
-- reproduces the shipped C body bug for bug (ratios read one slot off, ring lengths without the factor d, ids past the
-- last ring never written: SURVEY A10) for callers that want the numbers the unpatched script printed.
local dfe = require 'dfe_ffi'
local ffi = require 'ffi'
local bufs = {x = dfe.newBuffer(), y = dfe.newBuffer(), xo = dfe.newBuffer()}

function x2yxMulti2(geometry, x, compat_c)
   local ratios = ffi.new('int[?]', #geometry.ratios, geometry.ratios)
   local n = x:nElement()
   local xl = x
   if torch.typename(x) ~= 'torch.LongTensor' then xl = x:long() end
   local retx = torch.LongTensor():resizeAs(xl)
   local rety = torch.LongTensor():resizeAs(xl)
   local dx = dfe.upload(xl, bufs.x)
   local py, px = bufs.y:reserve(n * 8), bufs.xo:reserve(n * 8)
   dfe.check(dfe.lib.dfe_x2yx_multi(dfe.ctx, geometry.maxh, geometry.maxw, ratios, #geometry.ratios,
                                    ffi.cast('const int64_t*', dx), n, ffi.cast('int64_t*', py), ffi.cast('int64_t*', px),
                                    compat_c or 0))
   dfe.download(rety, py)
   dfe.download(retx, px)
   return rety, retx
end
