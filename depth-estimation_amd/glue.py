"""Thin glue modules of the reference with their own names: nn.SmartReshape, nn.FunctionWrapper, nn.Mul2,
nn.Log2, nn.OutputExtractor (SmartReshape.lua, FunctionWrapper.lua, Mul2.lua, Log.lua, OutputExtractor.lua),
plus postProcessImage (opticalflow_model.lua:323-472) and enlargeMask (depth_estimation_api.lua:76-132).
SmartReshape / FunctionWrapper / Mul2 / Log2 are shape book-keeping or one scalar elementwise op on a tensor
and are expressed with tensor views / in-place tensor ops; everything that reduces or filters goes to libdfe."""
import torch

from ._lib import lib
from .context import get_ctx, ptr
from .nn import Module


class SmartReshape(Module):
    """nn.SmartReshape(...): sizes are numbers (>= 0 literal, < 0 = size of input dim -n) or tables of such
    (product).  SmartReshape.lua:3-62."""

    def __init__(self, *sizes):
        super().__init__()
        self.sizes = list(sizes)

    @staticmethod
    def _size(code, input):
        if isinstance(code, (list, tuple)):
            r = 1
            for c in code:
                r *= SmartReshape._size(c, input)
            return r
        return code if code >= 0 else input.shape[-code - 1]

    def updateOutput(self, input):
        input = input.contiguous()  # :53
        size = [self._size(c, input) for c in self.sizes]
        n = 1
        for v in size:
            n *= v
        if n != input.numel():
            raise ValueError("SmartReshape: number of elements don't match.\n  input:size()=\n%s\n  self.sizes=\n%s" % (tuple(input.shape), self.sizes))  # :55-58
        self.output = input.view(size)
        return self.output

    def updateGradInput(self, input, gradOutput):
        self.gradInput = gradOutput.contiguous().view(input.shape)   # SmartReshape.lua:64-68
        return self.gradInput


class FunctionWrapper(Module):
    """nn.FunctionWrapper(init, updateOutput, updateGradInput): FunctionWrapper.lua:8-22"""

    def __init__(self, init, updateOutput, updateGradInput=None):
        super().__init__()
        init(self)
        self.fn_updateOutput = updateOutput
        self.fn_updateGradInput = updateGradInput

    def updateOutput(self, input):
        self.output = self.fn_updateOutput(self, input)
        return self.output


class Mul2(Module):
    """nn.Mul2: output = input * weight[1].  Mul2.lua:28-33"""

    def __init__(self):
        super().__init__()
        self.weight = torch.empty(1).uniform_(-1.0, 1.0)  # reset(): stdv = 1/sqrt(1)  :16-26

    def updateOutput(self, input):
        self.output = input * float(self.weight[0])
        return self.output


class Log2(Module):
    """nn.Log2(null_epsilon): clamps the INPUT in place to >= null_epsilon, then log.  Log.lua:13-22"""

    def __init__(self, null_epsilon=None):
        super().__init__()
        self.null_epsilon = null_epsilon

    def updateOutput(self, input):
        if input.dtype != torch.float32 or not input.is_contiguous():
            raise TypeError("Log2: input must be a contiguous FloatTensor (it is clamped in place, Log.lua:17)")
        out = torch.empty_like(input)
        ctx = get_ctx(input)
        # bad*eps + (1-bad)*input written back into input (:15-18), then log (:19-21)
        ctx.check(lib().dfe_log2_forward_f32(ctx.handle, ptr(input), input.numel(), float(self.null_epsilon or 0.0), int(self.null_epsilon is not None), ptr(out)))
        self.output = out
        return out

    def updateGradInput(self, input, gradOutput):
        go = gradOutput.contiguous()
        gi = torch.empty_like(go)
        ctx = get_ctx(go)
        ctx.check(lib().dfe_log2_backward_f32(ctx.handle, ptr(input), ptr(go), go.numel(), ptr(gi)))   # gradOutput / input (:24-28)
        self.gradInput = gi
        return gi


class OutputExtractor(Module):
    """nn.OutputExtractor(maxh, maxw): forward(input H x W x (maxh*maxw)) -> {x, y} soft arg-max in 1-based cell
    coordinates.  OutputExtractor.lua:3-35"""

    def __init__(self, maxh, maxw):
        super().__init__()
        self.maxh, self.maxw = int(maxh), int(maxw)

    def updateOutput(self, input):
        input = input.contiguous()
        if input.dtype != torch.float32 or input.shape[-1] != self.maxh * self.maxw:
            raise ValueError("OutputExtractor: input must be a FloatTensor ... x (maxh*maxw)")
        shape = input.shape[:-1]
        x = torch.empty(shape, dtype=torch.float32, device=input.device)
        y = torch.empty(shape, dtype=torch.float32, device=input.device)
        ctx = get_ctx(input)
        ctx.check(lib().dfe_output_extractor_f32(ctx.handle, ptr(input), x.numel(), self.maxh, self.maxw, ptr(x), ptr(y)))
        self.output = [x, y]
        return self.output


def postProcessImage(input, mask, winsize, method):
    """opticalflow_model.lua:323-472: method 'max' = masked mode filter of the rounded flow, anything else = masked
    per-component median.  input 2 x H x W, mask H x W -> 2 x H x W."""
    input, mask = input.contiguous(), mask.contiguous()
    _, H, W = input.shape
    out = torch.empty_like(input)
    ctx = get_ctx(input)
    ctx.check(lib().dfe_postprocess_image_f32(ctx.handle, ptr(input), ptr(mask), H, W, int(winsize), 0 if method == "max" else 1, ptr(out)))
    return out


def enlargeMask(mask, ix, iy):
    """depth_estimation_api.lua:76-132: erodes the validity mask by ix / iy pixels from each side of every row /
    column, in place; returns the mask."""
    if not mask.is_contiguous() or mask.dtype != torch.float32:
        raise TypeError("enlargeMask: contiguous FloatTensor expected")
    H, W = mask.shape
    ctx = get_ctx(mask)
    ctx.check(lib().dfe_enlarge_mask_f32(ctx.handle, ptr(mask), H, W, int(ix), int(iy)))
    return mask
