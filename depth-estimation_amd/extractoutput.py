"""`require 'extractoutput'` stand-in: same two functions, same argument order, results written
in place into caller tensors (extract_output.cpp:63-155, :157-255, registration :357-366)."""
import torch

from ._lib import lib
from .context import get_ctx, ptr


def _check(input, *outs):
    if input.dtype != torch.float32:
        raise TypeError("extractoutput: input must be a FloatTensor")  # hard-typed, extract_output.cpp:10-11
    if input.dim() != 3:
        raise ValueError("extractoutput: input must be H x W x N")
    for o in outs:
        if not o.is_contiguous() or tuple(o.shape) != tuple(input.shape[:2]):
            raise ValueError("extractoutput: outputs must be contiguous H x W tensors")


def extractOutput(input, scores, threshold, imaxs):
    """extractoutput.extractOutput(input HxWxN Float, scores HxW Float, threshold, imaxs HxW Long)"""
    _check(input, scores, imaxs)
    if scores.dtype != torch.float32 or imaxs.dtype != torch.int64:
        raise TypeError("extractOutput: scores must be Float and imaxs Long")
    input = input.contiguous()  # extract_output.cpp:71 newContiguous
    H, W, N = input.shape
    ctx = get_ctx(input)
    ctx.check(lib().dfe_extract_output(ctx.handle, ptr(input), H, W, N, ptr(scores), float(threshold), ptr(imaxs)))


def extractOutputMarginalized(input, threshold, threshold_acc, ret, retgd):
    """extractoutput.extractOutputMarginalized(input, threshold, threshold_acc, ret Long, retgd Long)"""
    _check(input, ret, retgd)
    if ret.dtype != torch.int64 or retgd.dtype != torch.int64:
        raise TypeError("extractOutputMarginalized: ret and retgd must be Long")
    input = input.contiguous()
    H, W, N = input.shape
    ctx = get_ctx(input)
    ctx.check(
        lib().dfe_extract_output_marginalized(
            ctx.handle, ptr(input), H, W, N, float(threshold), float(threshold_acc), ptr(ret), ptr(retgd)
        )
    )
