-- version2/test.lua:43-51 for one frame pair in ONE call on the MI355X path (UNTESTED here: no Lua runtime in the build image).
--   local Version2Flow = require 'Version2Flow'
--   local xflow, yflow = Version2Flow.flow(network, datap, prev, cur)     -- network = getNetwork(datap) of version2/network.lua
-- replaces:  local output = network:forward(input); local _, idx = output:min(3); idx = idx:add(-1):squeeze()
--            local yflow = (idx/datap.wWin):floor(); local xflow = idx-yflow*datap.wWin-datap.lWin; yflow = yflow-datap.tWin
-- The script's own `network` stays what it is (its modules keep working one by one through SpatialMatching.lua etc.); this entry reads
-- the normalisation kernel and the convolution weights out of network.modules[1].modules[1] -- where network:getWeights() finds
-- 'layer1' (version2/network.lua:32-36) -- and hands the whole pair to dfe_version2_flow_pair_f32: both normalisations, the crop, the
-- shared convolution stack, SpatialMatching(hWin, wWin) and the first-minimum decode run on the device, no volume comes back.
-- Uses setOption of dfe_ffi for the library's switches (dfe_set_option).
local dfe = require 'dfe_ffi'
local ffi = require 'ffi'
local M = {}
local B = {prev = dfe.newBuffer(), cur = dfe.newBuffer(), xf = dfe.newBuffer(), yf = dfe.newBuffer(), par = {}}

function M.setOption(key, value) dfe.check(dfe.lib.dfe_set_option(dfe.ctx, key, value)) end

function M.flow(network, datap, prev, cur)
   dfe.checktype(prev, 'torch.FloatTensor', 'Version2Flow.flow: prev')
   dfe.checktype(cur, 'torch.FloatTensor', 'Version2Flow.flow: cur')
   local C, H, W = prev:size(1), prev:size(2), prev:size(3)
   if cur:size(1) ~= C or cur:size(2) ~= H or cur:size(3) ~= W then error('Version2Flow.flow: frames of different sizes') end
   local filter1 = network.modules[1].modules[1]
   local scn = filter1.modules[1]                                   -- nn.SpatialContrastiveNormalization(3, image.gaussian1D(k))
   local kernel = scn.kernel:float():contiguous()
   if kernel:nDimension() ~= 1 then error('Version2Flow.flow: 1-D normalisation kernel expected (image.gaussian1D)') end
   local convs = {}
   for i = 3, #filter1.modules do                                   -- [1] normalisation, [2] SpatialPadding (crop), [3..] convolutions
      if filter1.modules[i].weight then table.insert(convs, filter1.modules[i]) end
   end
   local layers = ffi.new('dfe_filter_layer[?]', #convs)
   local hk, wk = 1, 1
   for i, m in ipairs(convs) do
      local L = layers[i - 1]
      B.par[i] = B.par[i] or {w = dfe.newBuffer(), b = dfe.newBuffer()}
      L.nIn, L.nOut, L.kH, L.kW = m.nInputPlane, m.nOutputPlane, m.kH, m.kW
      L.weight = ffi.cast('const float*', (dfe.upload(m.weight, B.par[i].w)))
      L.bias = ffi.cast('const float*', (dfe.upload(m.bias, B.par[i].b)))
      L.conn = nil; L.nConn = 0; L.tanh_after = 0
      hk, wk = hk + m.kH - 1, wk + m.kW - 1
   end
   local H1, W1 = H - (datap.hWin - 1) - (hk - 1), W - (datap.wWin - 1) - (wk - 1)
   if H1 < 1 or W1 < 1 then error('Version2Flow.flow: frame too small for the window and the kernels') end
   local dp, dc = dfe.upload(prev, B.prev), dfe.upload(cur, B.cur)
   local dxf = ffi.cast('float*', B.xf:reserve(H1 * W1 * 4))
   local dyf = ffi.cast('float*', B.yf:reserve(H1 * W1 * 4))
   dfe.check(dfe.lib.dfe_version2_flow_pair_f32(dfe.ctx, ffi.cast('const float*', dp), ffi.cast('const float*', dc), C, H, W, kernel:data(), kernel:size(1),
                                                scn.threshold or 1e-4, scn.thresval or 1e-4, layers, #convs, datap.hWin, datap.wWin, dxf, dyf, nil, nil))
   local xflow, yflow = torch.FloatTensor(H1, W1), torch.FloatTensor(H1, W1)
   dfe.download(xflow, dxf); dfe.download(yflow, dyf)
   return xflow:long(), yflow:long()                                -- (the script's xflow / yflow are LongTensors)
end

return M
