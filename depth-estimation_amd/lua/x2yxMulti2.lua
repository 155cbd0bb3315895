-- Drop-in for the global x2yxMulti2(geometry, x) of opticalflow_model_multiscale.lua:72-81 (which compiles x2yxMulti2.c
-- through `inline` at every call): class ids -> (y, x) displacements on the device.  UNTESTED here (no Lua runtime in the
-- build image).  compat_c = 1 reproduces the shipped C body bug for bug; pass 0 for the Lua scalar semantics that the
-- reference's own round-trip test (x2yxMultiNumber / yx2xMulti) pins.
local dfe = require 'dfe_ffi'
local ffi = require 'ffi'

function x2yxMulti2(geometry, x, compat_c)
   local ratios = ffi.new('int[?]', #geometry.ratios, geometry.ratios)
   local n = x:nElement()
   local retx = torch.LongTensor():resizeAs(x)
   local rety = torch.LongTensor():resizeAs(x)
   local dx = dfe.upload(x:long())
   local py, px = ffi.new('void*[1]'), ffi.new('void*[1]')
   dfe.check(dfe.lib.dfe_malloc(dfe.ctx, n * 8, py))
   dfe.check(dfe.lib.dfe_malloc(dfe.ctx, n * 8, px))
   dfe.check(dfe.lib.dfe_x2yx_multi(dfe.ctx, geometry.maxh, geometry.maxw, ratios, #geometry.ratios,
                                    ffi.cast('const int64_t*', dx), n, ffi.cast('int64_t*', py[0]), ffi.cast('int64_t*', px[0]),
                                    compat_c == nil and 1 or compat_c))
   dfe.download(rety, py[0])
   dfe.download(retx, px[0])
   dfe.free(dx); dfe.free(py[0]); dfe.free(px[0])
   return rety, retx
end
