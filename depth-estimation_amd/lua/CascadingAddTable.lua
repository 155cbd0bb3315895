-- Drop-in for the reference's nn.CascadingAddTable on the MI355X path (UNTESTED here: no Lua runtime in the build image).
-- Same constructor (ratios, trainable, single_beta), same updateOutput(table of (H*W) x Kh x Kw tensors) -> table of the
-- same shapes, same error texts for the shape checks (CascadingAddTable.lua:108-135); updateGradInput follows :137-154.
-- HEAD's graph has no trainable parameter (the Mul2 gains and the Power normaliser are commented out, :29,46,57,61), so
-- `trainable` / `single_beta` are accepted and kept, like the reference's constructor and the Python mirror do
-- (tests/test_cascad.lua:14 constructs the module with trainable = true), and accGradParameters stays a no-op.
local dfe = require 'dfe_ffi'
local ffi = require 'ffi'
local CascadingAddTable, parent = torch.class('nn.CascadingAddTable', 'nn.Module')

function CascadingAddTable:__init(ratios, trainable, single_beta)
   parent.__init(self)
   self.ratios = ratios
   if trainable == nil then self.trainable = true else self.trainable = trainable end   -- (the reference's default, CascadingAddTable.lua:11)
   self.single_beta = single_beta or false
   self.output = {}
   self.gradInput = {}
   for i = 1, #ratios do
      self.output[i] = torch.FloatTensor()
      self.gradInput[i] = torch.FloatTensor()
   end
   self.cratios = ffi.new('int[?]', #ratios, ratios)
   self.dbuf = {}
end

local function check_inputs(self, input)
   for i = 1, #input do
      if input[i]:nDimension() ~= 3 then
         error('nn.CascadingAddTable: input must be a table of 3D-tensors (HxW) x Kh x Kw')
      end
   end
   if #input ~= #self.ratios then
      error('nn.CascadingAddTable: input and ratios must have the same size')
   end
   for i = 1, #input - 1 do
      local r, r2 = self.ratios[i], self.ratios[i + 1]
      if (math.fmod(input[i]:size(2) * (r2 - r), 2 * r2) ~= 0) or (math.fmod(input[i]:size(3) * (r2 - r), 2 * r2) ~= 0) then
         error('nn.CascadingAddTable: ratios and input sizes not compatible')
      end
   end
end

-- runs one of the two C entry points over a table of tensors: stage into the module's persistent device buffers, call, fetch
local function run(self, fn, src, dst)
   local n = #src
   local P, maxh, maxw = src[1]:size(1), src[1]:size(2), src[1]:size(3)
   local din, dout = ffi.new('const float*[?]', n), ffi.new('float*[?]', n)
   for i = 1, n do
      dfe.checktype(src[i], 'torch.FloatTensor', 'nn.CascadingAddTable: input[' .. i .. ']')
      self.dbuf[i] = self.dbuf[i] or {dfe.newBuffer(), dfe.newBuffer()}
      din[i - 1] = ffi.cast('const float*', (dfe.upload(src[i], self.dbuf[i][1])))
      dout[i - 1] = ffi.cast('float*', self.dbuf[i][2]:reserve(src[i]:nElement() * 4))
   end
   dfe.check(fn(dfe.ctx, din, self.cratios, n, P, maxh, maxw, dout))
   for i = 1, n do
      dst[i]:resizeAs(src[i])
      dfe.download(dst[i], dout[i - 1])
   end
end

function CascadingAddTable:updateOutput(input)
   check_inputs(self, input)
   run(self, dfe.lib.dfe_cascading_add_f32, input, self.output)
   return self.output
end

function CascadingAddTable:updateGradInput(input, gradOutput)
   run(self, dfe.lib.dfe_cascading_add_backward_f32, gradOutput, self.gradInput)
   return self.gradInput
end

function CascadingAddTable:accGradParameters(input, gradOutput, scale)
end
