-- `require 'extractoutput'` drop-in (UNTESTED here): same two functions and argument order as
-- extract_output.cpp:357-366, results written in place into the caller's tensors.  Device buffers persist between calls.
local dfe = require 'dfe_ffi'
extractoutput = {}
local b = {dfe.newBuffer(), dfe.newBuffer(), dfe.newBuffer()}

function extractoutput.extractOutput(input, scores, threshold, imaxs)
   dfe.checktype(input, 'torch.FloatTensor', 'extractOutput: input')     -- extract_output.cpp:10-11 is hard-typed to THFloatTensor
   dfe.checktype(scores, 'torch.FloatTensor', 'extractOutput: scores')
   dfe.checktype(imaxs, 'torch.LongTensor', 'extractOutput: imaxs')
   local H, W, N = input:size(1), input:size(2), input:size(3)
   local din, ds, di = dfe.upload(input, b[1]), dfe.upload(scores, b[2]), dfe.upload(imaxs, b[3])
   dfe.check(dfe.lib.dfe_extract_output(dfe.ctx, din, H, W, N, ds, threshold, di))
   dfe.download(scores, ds); dfe.download(imaxs, di)
end

function extractoutput.extractOutputMarginalized(input, threshold, threshold_acc, ret, retgd)
   dfe.checktype(input, 'torch.FloatTensor', 'extractOutputMarginalized: input')
   dfe.checktype(ret, 'torch.LongTensor', 'extractOutputMarginalized: ret')
   dfe.checktype(retgd, 'torch.LongTensor', 'extractOutputMarginalized: retgd')
   local H, W, N = input:size(1), input:size(2), input:size(3)
   local din, dr, dg = dfe.upload(input, b[1]), dfe.upload(ret, b[2]), dfe.upload(retgd, b[3])
   dfe.check(dfe.lib.dfe_extract_output_marginalized(dfe.ctx, din, H, W, N, threshold, threshold_acc, dr, dg))
   dfe.download(ret, dr); dfe.download(retgd, dg)
end

return extractoutput
