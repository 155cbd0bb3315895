-- Drop-ins for the modules of getFilter (opticalflow_model.lua:45-79, radial/radial_opticalflow_network.lua:6-30) on the MI355X
-- path: nn.SpatialConvolution, nn.SpatialConvolutionMap and nn.Tanh keep their classes, constructors, parameters
-- (weight / bias / gradWeight / gradBias stay Torch FloatTensors, so :clone('weight', ...), getWeights, saveModel and the
-- optimiser see what they always saw) -- only updateOutput / updateGradInput / accGradParameters are re-pointed at libdfe.
-- UNTESTED here (no Lua runtime in the build image); the same entry points run under tests/ through the ctypes binding.
--   require 'FilterStack'            -- after require 'nn': patches the three classes in place
local dfe = require 'dfe_ffi'
local ffi = require 'ffi'

local function bufs(self)
   if not self._dfe then
      self._dfe = {x = dfe.newBuffer(), w = dfe.newBuffer(), b = dfe.newBuffer(), y = dfe.newBuffer(), g = dfe.newBuffer(),
                   gi = dfe.newBuffer(), gw = dfe.newBuffer(), gb = dfe.newBuffer(), c = dfe.newBuffer()}
   end
   return self._dfe
end

local function check3d(t, who)
   dfe.checktype(t, 'torch.FloatTensor', who)
   if t:nDimension() ~= 3 then error(who .. ' must be nInputPlane x H x W') end
end

-- nn.SpatialConvolution(nIn, nOut, kW, kH): weight nOut x nIn x kH x kW (the layout dfe_spatial_convolution_f32 reads)
local Conv = nn.SpatialConvolution
function Conv:updateOutput(input)
   check3d(input, 'nn.SpatialConvolution: input')
   local b = bufs(self)
   local nIn, H, W = input:size(1), input:size(2), input:size(3)
   if nIn ~= self.nInputPlane then error('nn.SpatialConvolution: wrong number of input planes') end
   self.output:resize(self.nOutputPlane, H - self.kH + 1, W - self.kW + 1)
   local dx, dw, db = dfe.upload(input, b.x), dfe.upload(self.weight, b.w), dfe.upload(self.bias, b.b)
   local dy = b.y:reserve(self.output:nElement() * 4)
   if self.kernel == 'mfma' then   -- opt-in: matrix cores (fused multiply-adds in the reference's k order)
      dfe.check(dfe.lib.dfe_spatial_convolution_mfma_f32(dfe.ctx, dx, dw, db, nIn, self.nOutputPlane, H, W, self.kH, self.kW, 0, dy))
   else
      dfe.check(dfe.lib.dfe_spatial_convolution_f32(dfe.ctx, dx, dw, db, nIn, self.nOutputPlane, H, W, self.kH, self.kW, dy))
   end
   dfe.download(self.output, dy)
   return self.output
end
function Conv:updateGradInput(input, gradOutput)
   check3d(input, 'nn.SpatialConvolution: input'); check3d(gradOutput, 'nn.SpatialConvolution: gradOutput')
   local b = bufs(self)
   local nIn, H, W = input:size(1), input:size(2), input:size(3)
   self.gradInput:resizeAs(input)
   local dg, dw = dfe.upload(gradOutput, b.g), dfe.upload(self.weight, b.w)
   local dgi = b.gi:reserve(input:nElement() * 4)
   dfe.check(dfe.lib.dfe_spatial_convolution_grad_input_f32(dfe.ctx, dg, dw, nIn, self.nOutputPlane, H, W, self.kH, self.kW, dgi))
   dfe.download(self.gradInput, dgi)
   return self.gradInput
end
function Conv:accGradParameters(input, gradOutput, scale)
   local b = bufs(self)
   local nIn, H, W = input:size(1), input:size(2), input:size(3)
   local dx, dg = dfe.upload(input, b.x), dfe.upload(gradOutput, b.g)
   local dgw, dgb = dfe.upload(self.gradWeight, b.gw), dfe.upload(self.gradBias, b.gb)     -- ACCUMULATES into the shared buffers
   dfe.check(dfe.lib.dfe_spatial_convolution_acc_grad_f32(dfe.ctx, dx, dg, nIn, self.nOutputPlane, H, W, self.kH, self.kW, scale or 1, dgw, dgb))
   dfe.download(self.gradWeight, dgw); dfe.download(self.gradBias, dgb)
end

-- nn.SpatialConvolutionMap(connTable, kW, kH): weight nConn x kH x kW, connTable nConn x 2 = (from, to), 1-based
local ConvMap = nn.SpatialConvolutionMap
local function conn_i32(self, b)
   local t = self.connTable:int():contiguous()
   return ffi.cast('const int32_t*', (dfe.upload(t, b.c))), self.connTable:size(1)
end
function ConvMap:updateOutput(input)
   check3d(input, 'nn.SpatialConvolutionMap: input')
   local b = bufs(self)
   local nIn, H, W = input:size(1), input:size(2), input:size(3)
   self.output:resize(self.nOutputPlane, H - self.kH + 1, W - self.kW + 1)
   local dx, dw, db = dfe.upload(input, b.x), dfe.upload(self.weight, b.w), dfe.upload(self.bias, b.b)
   local dc, nConn = conn_i32(self, b)
   local dy = b.y:reserve(self.output:nElement() * 4)
   dfe.check(dfe.lib.dfe_spatial_convolution_map_f32(dfe.ctx, dx, dw, db, dc, nConn, nIn, self.nOutputPlane, H, W, self.kH, self.kW, dy))
   dfe.download(self.output, dy)
   return self.output
end
function ConvMap:updateGradInput(input, gradOutput)
   local b = bufs(self)
   local nIn, H, W = input:size(1), input:size(2), input:size(3)
   self.gradInput:resizeAs(input)
   local dg, dw = dfe.upload(gradOutput, b.g), dfe.upload(self.weight, b.w)
   local dc, nConn = conn_i32(self, b)
   local dgi = b.gi:reserve(input:nElement() * 4)
   dfe.check(dfe.lib.dfe_spatial_convolution_map_grad_input_f32(dfe.ctx, dg, dw, dc, nConn, nIn, self.nOutputPlane, H, W, self.kH, self.kW, dgi))
   dfe.download(self.gradInput, dgi)
   return self.gradInput
end
function ConvMap:accGradParameters(input, gradOutput, scale)
   local b = bufs(self)
   local nIn, H, W = input:size(1), input:size(2), input:size(3)
   local dx, dg = dfe.upload(input, b.x), dfe.upload(gradOutput, b.g)
   local dgw, dgb = dfe.upload(self.gradWeight, b.gw), dfe.upload(self.gradBias, b.gb)
   local dc, nConn = conn_i32(self, b)
   dfe.check(dfe.lib.dfe_spatial_convolution_map_acc_grad_f32(dfe.ctx, dx, dg, dc, nConn, nIn, self.nOutputPlane, H, W, self.kH, self.kW, scale or 1, dgw, dgb))
   dfe.download(self.gradWeight, dgw); dfe.download(self.gradBias, dgb)
end

-- nn.Tanh
local Tanh = nn.Tanh
function Tanh:updateOutput(input)
   dfe.checktype(input, 'torch.FloatTensor', 'nn.Tanh: input')
   local b = bufs(self)
   self.output:resizeAs(input)
   local dx = dfe.upload(input, b.x)
   local dy = b.y:reserve(input:nElement() * 4)
   dfe.check(dfe.lib.dfe_tanh_f32(dfe.ctx, dx, input:nElement(), dy))
   dfe.download(self.output, dy)
   return self.output
end
function Tanh:updateGradInput(input, gradOutput)
   local b = bufs(self)
   self.gradInput:resizeAs(gradOutput)
   local dy, dg = dfe.upload(self.output, b.y), dfe.upload(gradOutput, b.g)
   local dgi = b.gi:reserve(gradOutput:nElement() * 4)
   dfe.check(dfe.lib.dfe_tanh_backward_f32(dfe.ctx, dy, dg, gradOutput:nElement(), dgi))
   dfe.download(self.gradInput, dgi)
   return self.gradInput
end

return {SpatialConvolution = Conv, SpatialConvolutionMap = ConvMap, Tanh = Tanh}
