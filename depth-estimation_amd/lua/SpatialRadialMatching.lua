-- Drop-in for nnx's nn.SpatialRadialMatching(hWin) on the MI355X path (UNTESTED here: no Lua runtime in the build
-- image).  forward({input1 K x H1 x W, input2 K x (H1+hWin-1) x W}) -> H1 x W x hWin, the contract of the call sites
-- radial/radial_opticalflow_network.lua:33,71-72 and radial/radial_opticalflow_groundtruth.lua:152.
-- self.flow (H1 x W, = output:min(3) - 1 as the callers compute it, train_radial:166-167) is filled on the way when the
-- window has a fused instantiation (hWin 8, 12, 15, 16).
local dfe = require 'dfe_ffi'
local SpatialRadialMatching, parent = torch.class('nn.SpatialRadialMatching', 'nn.Module')

function SpatialRadialMatching:__init(hWin)
   parent.__init(self)
   self.hWin = hWin
   self.flow = torch.FloatTensor()
   self.d1, self.d2, self.dout, self.dflow = dfe.newBuffer(), dfe.newBuffer(), dfe.newBuffer(), dfe.newBuffer()
end

function SpatialRadialMatching:updateOutput(input)
   local in1, in2 = input[1], input[2]
   dfe.checktype(in1, 'torch.FloatTensor', 'nn.SpatialRadialMatching: input[1]')
   dfe.checktype(in2, 'torch.FloatTensor', 'nn.SpatialRadialMatching: input[2]')
   local K, H1, W = in1:size(1), in1:size(2), in1:size(3)
   if in2:size(1) ~= K or in2:size(2) ~= H1 + self.hWin - 1 or in2:size(3) ~= W then
      error('nn.SpatialRadialMatching: input[2] must be K x (H1+hWin-1) x W')
   end
   self.output:resize(H1, W, self.hWin)
   local d1, d2 = dfe.upload(in1, self.d1), dfe.upload(in2, self.d2)
   local dout = self.dout:reserve(self.output:nElement() * 4)
   local fused = self.hWin == 8 or self.hWin == 12 or self.hWin == 15 or self.hWin == 16
   if fused then
      self.flow:resize(H1, W)
      local dflow = self.dflow:reserve(H1 * W * 4)
      dfe.check(dfe.lib.dfe_radial_match_argmin_f32(dfe.ctx, d1, H1, d2, K, H1, W, self.hWin, dout, dflow, 0))
      dfe.download(self.flow, dflow)
   else
      dfe.check(dfe.lib.dfe_radial_matching_f32(dfe.ctx, d1, d2, K, H1, W, self.hWin, dout))
   end
   dfe.download(self.output, dout)
   return self.output
end
