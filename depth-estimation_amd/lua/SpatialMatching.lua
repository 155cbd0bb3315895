-- Drop-in for nnx's nn.SpatialMatching(maxh, maxw, false) on the MI355X path (UNTESTED here: no Lua
-- runtime in the build image).  Same constructor, same forward({input1, input2}) -> H1 x W1 x maxh x maxw
-- contract as the call sites opticalflow_model.lua:93 and opticalflow_model_multiscale.lua:216.
-- Device buffers live with the module (allocated on resize, not per call); inputs / output are pinned in place.
local dfe = require 'dfe_ffi'
local SpatialMatching, parent = torch.class('nn.SpatialMatching', 'nn.Module')

function SpatialMatching:__init(maxh, maxw, full_output)
   parent.__init(self)
   assert(not full_output, 'nn.SpatialMatching (dfe): only full_output=false is used by the reference')
   self.maxh, self.maxw = maxh, maxw
   self.d1, self.d2, self.dout = dfe.newBuffer(), dfe.newBuffer(), dfe.newBuffer()
end

function SpatialMatching:updateOutput(input)
   local in1, in2 = input[1], input[2]
   dfe.checktype(in1, 'torch.FloatTensor', 'nn.SpatialMatching: input[1]')
   dfe.checktype(in2, 'torch.FloatTensor', 'nn.SpatialMatching: input[2]')
   local K, H1, W1 = in1:size(1), in1:size(2), in1:size(3)
   if in2:size(1) ~= K or in2:size(2) ~= H1 + self.maxh - 1 or in2:size(3) ~= W1 + self.maxw - 1 then
      error('nn.SpatialMatching: input[2] must be K x (H1+maxh-1) x (W1+maxw-1)')
   end
   self.output:resize(H1, W1, self.maxh, self.maxw)
   local d1, d2 = dfe.upload(in1, self.d1), dfe.upload(in2, self.d2)
   local dout = self.dout:reserve(self.output:nElement() * 4)
   dfe.check(dfe.lib.dfe_spatial_matching_f32(dfe.ctx, d1, d2, K, H1, W1, self.maxh, self.maxw, dout))
   dfe.download(self.output, dout)
   return self.output
end
