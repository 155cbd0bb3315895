-- `require 'extractoutput'` drop-in (UNTESTED here): same two functions and argument order as
-- extract_output.cpp:357-366, results written in place into the caller's tensors.
local dfe = require 'dfe_ffi'
local ffi = require 'ffi'
extractoutput = {}

function extractoutput.extractOutput(input, scores, threshold, imaxs)
   assert(torch.typename(input) == 'torch.FloatTensor', 'extractOutput: FloatTensor expected')  -- extract_output.cpp:10-11
   local H, W, N = input:size(1), input:size(2), input:size(3)
   local din, ds, di = dfe.upload(input), dfe.upload(scores), dfe.upload(imaxs)
   dfe.check(dfe.lib.dfe_extract_output(dfe.ctx, din, H, W, N, ds, threshold, di))
   dfe.download(scores, ds); dfe.download(imaxs, di)
   dfe.free(din); dfe.free(ds); dfe.free(di)
end

function extractoutput.extractOutputMarginalized(input, threshold, threshold_acc, ret, retgd)
   local H, W, N = input:size(1), input:size(2), input:size(3)
   local din, dr, dg = dfe.upload(input), dfe.upload(ret), dfe.upload(retgd)
   dfe.check(dfe.lib.dfe_extract_output_marginalized(dfe.ctx, din, H, W, N, threshold, threshold_acc, dr, dg))
   dfe.download(ret, dr); dfe.download(retgd, dg)
   dfe.free(din); dfe.free(dr); dfe.free(dg)
end

return extractoutput
