-- depth_estimation_opticalflow.lua:103-116 for a single-scale model in ONE call on the MI355X path (UNTESTED here: no Lua runtime in the
-- build image).
--   local SingleScaleFlow = require 'SingleScaleFlow'
--   local output = SingleScaleFlow.forwardFlow(geometry, filter, last_frame, frame [, threshold])   -- frames C x H x W FloatTensors
--   -- or, with each frame's features kept for the next pair as the script does (geometry.prefilter):
--   local output = SingleScaleFlow.forwardFlowPrefiltered(geometry, last_im, im [, threshold])       -- feature maps K x Hf x Wf
-- replaces:  local input = prepareInput(geometry, last_im, im)          opticalflow_model.lua:131-151 (the narrow of patch 1)
--            local moutput = model:forward(input)                       getModel(geometry, true, true): SpatialMatching -> Minus -> SoftMax (:81-129)
--            local output = processOutput(geometry, moutput, true [, threshold])   (:201-252)
-- and returns processOutput's table: output.full (2 x hImg x wImg, plane 1 = y, plane 2 = x), output.full_confidences (hImg x wImg),
-- output.index (h x w LongTensor), output.y / output.x (h x w, centred displacements), output.confidences.  The filter's weights are read
-- from `filter` (getFilter(geometry)) on every call, so a model that keeps training keeps working; the script's own model:forward stays
-- what it is, module by module.  With 16- / 17-wide windows the H1 x W1 x maxh x maxw volume is never written (dfe_flow_pair_filtered_f32).
local dfe = require 'dfe_ffi'
local ffi = require 'ffi'
local M = {}
local B = {a = dfe.newBuffer(), b = dfe.newBuffer(), full = dfe.newBuffer(), conf = dfe.newBuffer(), idx = dfe.newBuffer(), par = {}}

local function layers_of(filter)
   local convs = {}
   for i = 1, #filter.modules do
      local m = filter.modules[i]
      if m.weight then
         if m.connTable then error('SingleScaleFlow: nn.SpatialConvolutionMap layers go through MultiscaleFlow / FilterStack') end
         table.insert(convs, {m = m, tanh = (filter.modules[i + 1] and torch.typename(filter.modules[i + 1]) == 'nn.Tanh') and 1 or 0})
      end
   end
   local layers = ffi.new('dfe_filter_layer[?]', #convs)
   local hk, wk = 1, 1
   for i, c in ipairs(convs) do
      local L, m = layers[i - 1], c.m
      B.par[i] = B.par[i] or {w = dfe.newBuffer(), b = dfe.newBuffer()}
      L.nIn, L.nOut, L.kH, L.kW = m.nInputPlane, m.nOutputPlane, m.kH, m.kW
      L.weight = ffi.cast('const float*', (dfe.upload(m.weight, B.par[i].w)))
      L.bias = ffi.cast('const float*', (dfe.upload(m.bias, B.par[i].b)))
      L.conn = nil; L.nConn = 0; L.tanh_after = c.tanh
      hk, wk = hk + m.kH - 1, wk + m.kW - 1
   end
   return layers, #convs, hk, wk
end

local function run(geometry, layers, nlayers, hk, wk, a, b, threshold)
   dfe.checktype(a, 'torch.FloatTensor', 'SingleScaleFlow: patch 1')
   dfe.checktype(b, 'torch.FloatTensor', 'SingleScaleFlow: patch 2')
   local C, H, W = a:size(1), a:size(2), a:size(3)
   if b:size(1) ~= C or b:size(2) ~= H or b:size(3) ~= W then error('SingleScaleFlow: patches of different sizes') end   -- assert(sameSize(patch1, patch2))
   local h, w = H - hk + 1 - geometry.maxh + 1, W - wk + 1 - geometry.maxw + 1
   if h < 1 or w < 1 then error('SingleScaleFlow: frame too small for the window and the kernels') end
   local hImg, wImg = geometry.hImg, geometry.wImg
   local da, db = dfe.upload(a, B.a), dfe.upload(b, B.b)
   local dfull = ffi.cast('float*', B.full:reserve(2 * hImg * wImg * 4))
   local dconf = ffi.cast('float*', B.conf:reserve(hImg * wImg * 4))
   local didx = ffi.cast('int64_t*', B.idx:reserve(h * w * 8))
   dfe.check(dfe.lib.dfe_flow_pair_filtered_f32(dfe.ctx, ffi.cast('const float*', da), ffi.cast('const float*', db), C, H, W, layers, nlayers,
                                                geometry.maxh, geometry.maxw, threshold and 1 or 0, threshold or 0, hImg, wImg, dfull, dconf, didx, nil))
   local ret = {full = torch.FloatTensor(2, hImg, wImg), full_confidences = torch.FloatTensor(hImg, wImg), index = torch.LongTensor(h, w)}
   dfe.download(ret.full, dfull); dfe.download(ret.full_confidences, dconf); dfe.download(ret.index, didx)
   local ho, wo = math.floor((hImg - h) / 2), math.floor((wImg - w) / 2)                              -- opticalflow_model.lua:228-230
   ret.y = ret.full[1]:sub(1 + ho, h + ho, 1 + wo, w + wo):clone()
   ret.x = ret.full[2]:sub(1 + ho, h + ho, 1 + wo, w + wo):clone()
   ret.confidences = ret.full_confidences:sub(1 + ho, h + ho, 1 + wo, w + wo):clone()
   return ret
end

function M.forwardFlow(geometry, filter, patch1, patch2, threshold)
   local layers, n, hk, wk = layers_of(filter)
   return run(geometry, layers, n, hk, wk, patch1, patch2, threshold)
end

function M.forwardFlowPrefiltered(geometry, feat1, feat2, threshold)
   return run(geometry, nil, 0, 1, 1, feat1, feat2, threshold)
end

return M
