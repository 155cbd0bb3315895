-- Drop-ins for the inline-C post filters of the dense drivers (UNTESTED here: no Lua runtime in the build image):
--   postProcessImage(input, mask, winsize, method)   opticalflow_model.lua:323-472  ('max' = mode filter, else median)
--   enlargeMask(mask, ix, iy)                        depth_estimation_api.lua:76-132 (in place)
-- Same global names and argument order: `require 'postfilters'` after the reference's own files overrides them.
local dfe = require 'dfe_ffi'
local B = {flow = dfe.newBuffer(), mask = dfe.newBuffer(), out = dfe.newBuffer()}

function postProcessImage(input, mask, winsize, method)
   dfe.checktype(input, 'torch.FloatTensor', 'postProcessImage: input')
   dfe.checktype(mask, 'torch.FloatTensor', 'postProcessImage: mask')
   if input:nDimension() ~= 3 or input:size(1) ~= 2 then error('postProcessImage: input must be 2 x H x W (y-flow, x-flow)') end
   local H, W = input:size(2), input:size(3)
   local ret = torch.FloatTensor(2, H, W)
   local dflow, dmask = dfe.upload(input, B.flow), dfe.upload(mask, B.mask)
   local dout = B.out:reserve(ret:nElement() * 4)
   dfe.check(dfe.lib.dfe_postprocess_image_f32(dfe.ctx, dflow, dmask, H, W, winsize, (method == 'max') and 0 or 1, dout))
   dfe.download(ret, dout)
   return ret
end

function enlargeMask(mask, ix, iy)
   dfe.checktype(mask, 'torch.FloatTensor', 'enlargeMask: mask')
   assert(mask:isContiguous(), 'enlargeMask: mask must be contiguous (it is modified in place)')
   local dmask = dfe.upload(mask, B.mask)
   dfe.check(dfe.lib.dfe_enlarge_mask_f32(dfe.ctx, dmask, mask:size(1), mask:size(2), ix, iy))
   dfe.download(mask, dmask)
   return mask
end
