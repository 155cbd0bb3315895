-- Optional fast path for the per-frame body of the dense driver with a MULTISCALE model (depth_estimation_opticalflow.lua:113-116:
-- `moutput = model:forward(input); output = processOutput(geometry, moutput, true)`), raw-patch or with the learned filters of
-- getModelMultiscale (opticalflow_model_multiscale.lua:196-229): ONE C call, no per-scale Torch tensors.
--   local msflow = require 'MultiscaleFlow'
--   local out = msflow.forwardFlow(geometry, model, I0, I1)    -- {index = LongTensor HxW, y, x, full = 2 x H x W, confidences}
-- The weights are read from model:getWeights() on every call ('layer<i>' when geometry.share_filters, else
-- 'scale<r>_layer<i>', opticalflow_model_multiscale.lua:347-370) and the biases from the filter modules themselves
-- (getWeights lists no biases, opticalflow_model.lua:66-76), so a loadModel'ed or freshly trained model needs no conversion.
-- UNTESTED here (no Lua runtime in the build image); tests/ drive the same entry through MultiscaleModel.forwardFlow.
local dfe = require 'dfe_ffi'
local ffi = require 'ffi'
local M = {}
local B = {i0 = dfe.newBuffer(), i1 = dfe.newBuffer(), flow = dfe.newBuffer(), idx = dfe.newBuffer(), par = {}}

local function filter_of_scale(model, i)   -- processors[i].modules[1].modules[1].modules[3] (:353-363)
   return model.pyramid.processors[i].modules[1].modules[1].modules[3]
end

function M.forwardFlow(geometry, model, I0, I1)
   dfe.checktype(I0, 'torch.FloatTensor', 'forwardFlow: I0'); dfe.checktype(I1, 'torch.FloatTensor', 'forwardFlow: I1')
   local C, H, W = I0:size(1), I0:size(2), I0:size(3)
   local rmax = geometry.ratios[#geometry.ratios]
   if H % rmax ~= 0 or W % rmax ~= 0 then   -- opticalflow_model_multiscale.lua:234-248: zero-pad bottom / right to a multiple of rmax
      local th, tw = rmax * math.ceil(H / rmax), rmax * math.ceil(W / rmax)
      local p0, p1 = torch.FloatTensor(C, th, tw):zero(), torch.FloatTensor(C, th, tw):zero()
      p0:sub(1, C, 1, H, 1, W):copy(I0); p1:sub(1, C, 1, H, 1, W):copy(I1)
      I0, I1, H, W = p0, p1, th, tw
   end
   local nr = #geometry.ratios
   local ratios = ffi.new('int[?]', nr, geometry.ratios)
   local d0, d1 = dfe.upload(I0, B.i0), dfe.upload(I1, B.i1)
   local dflow = ffi.cast('float*', B.flow:reserve(2 * H * W * 4))
   local didx = ffi.cast('int64_t*', B.idx:reserve(H * W * 8))
   if geometry.layers and #geometry.layers > 0 then
      local nstacks = geometry.share_filters and 1 or nr
      local layers, nl, k = nil, 0, 0
      for s = 1, nstacks do
         local mods, prev = filter_of_scale(model, s).modules, nil
         local convs = {}
         for _, m in ipairs(mods) do
            if m.weight then table.insert(convs, {m = m, tanh = 0}); prev = convs[#convs]
            elseif torch.typename(m) == 'nn.Tanh' and prev then prev.tanh = 1 end
         end
         if not layers then nl = #convs; layers = ffi.new('dfe_filter_layer[?]', nstacks * nl) end
         for _, c in ipairs(convs) do
            local m, L = c.m, layers[k]
            B.par[k] = B.par[k] or {w = dfe.newBuffer(), b = dfe.newBuffer(), c = dfe.newBuffer()}
            L.nIn, L.nOut, L.kH, L.kW = m.nInputPlane, m.nOutputPlane, m.kH, m.kW
            L.weight = ffi.cast('const float*', (dfe.upload(m.weight, B.par[k].w)))
            L.bias = ffi.cast('const float*', (dfe.upload(m.bias, B.par[k].b)))
            if m.connTable then
               L.conn = ffi.cast('const int32_t*', (dfe.upload(m.connTable:int():contiguous(), B.par[k].c))); L.nConn = m.connTable:size(1)
            else
               L.conn = nil; L.nConn = 0
            end
            L.tanh_after = c.tanh
            k = k + 1
         end
      end
      dfe.check(dfe.lib.dfe_multiscale_flow_pair_filtered_f32(dfe.ctx, d0, d1, C, H, W, geometry.maxh, geometry.maxw, ratios, nr, layers, nl,
                                                             geometry.share_filters and 1 or 0, 0, dflow, didx))
   else
      dfe.check(dfe.lib.dfe_multiscale_flow_pair_f32(dfe.ctx, d0, d1, C, H, W, geometry.hKernel, geometry.maxh, geometry.maxw, ratios, nr, dflow, didx))
   end
   local full, index = torch.FloatTensor(2, H, W), torch.LongTensor(H, W)
   dfe.download(full, dflow); dfe.download(index, didx)
   return {index = index, y = full[1]:long(), x = full[2]:long(), full = full, confidences = torch.FloatTensor(H, W):fill(1),
           full_confidences = torch.FloatTensor(H, W):fill(1)}
end

return M
