"""Class-index codecs and output post-processing with the reference's names
(opticalflow_model.lua, opticalflow_model_multiscale.lua).  `geometry` is a dict (or any object
with the same attributes) holding the reference's fields: maxh, maxw, ratios, multiscale,
hImg, wImg, output_extraction_method."""
import ctypes as C
import math

import torch

from ._lib import lib, ratios_array, DfeError
from .context import get_ctx, ptr
from . import extractoutput


def _g(geometry, key, default=None):
    if isinstance(geometry, dict):
        return geometry.get(key, default)
    return getattr(geometry, key, default)


def yx2x(geometry, y, x):
    """opticalflow_model.lua:12-14"""
    return (y - 1) * _g(geometry, "maxw") + x


def centered2onebased(geometry, y, x):
    """opticalflow_model.lua:28-30"""
    return y + math.ceil(_g(geometry, "maxh") / 2), x + math.ceil(_g(geometry, "maxw") / 2)


def onebased2centered(geometry, y, x):
    """opticalflow_model.lua:31-33"""
    return y - math.ceil(_g(geometry, "maxh") / 2), x - math.ceil(_g(geometry, "maxw") / 2)


def x2yx(geometry, x):
    """opticalflow_model.lua:16-25: 1-based class id -> 1-based (ty, tx); number or LongTensor."""
    maxw = _g(geometry, "maxw")
    if isinstance(x, (int, float)):
        x = int(x)
        return (x - 1) // maxw + 1, (x - 1) % maxw + 1
    if x.dtype != torch.int64:
        raise TypeError("x2yx: LongTensor expected")
    x = x.contiguous()
    ctx = get_ctx(x)
    y = torch.empty_like(x)
    xo = torch.empty_like(x)
    maxh = _g(geometry, "maxh")
    # the device op returns centred displacements (ty - ceil(maxh/2)); undo the centring to keep
    # x2yx's 1-based contract: ty = y + floor((maxh-1)/2) + 1
    ctx.check(lib().dfe_x2yx(ctx.handle, ptr(x), x.numel(), maxh, maxw, ptr(y), ptr(xo)))
    return y + ((maxh - 1) // 2 + 1), xo + ((maxw - 1) // 2 + 1)


def _ratios(geometry):
    return ratios_array(_g(geometry, "ratios"))


def yx2xMulti(geometry, y, x):
    """opticalflow_model_multiscale.lua:10-52 (scalar)"""
    r, n = _ratios(geometry)
    v = lib().dfe_yx2x_multi(_g(geometry, "maxh"), _g(geometry, "maxw"), r, n, float(y), float(x))
    if v < 0:
        raise AssertionError("yx2xMulti: (%s,%s) is outside every scale's window" % (y, x))  # :29 assert
    return int(v)


def x2yxMultiNumber(geometry, x):
    """opticalflow_model_multiscale.lua:83-132"""
    r, n = _ratios(geometry)
    oy, ox = C.c_int64(), C.c_int64()
    rc = lib().dfe_x2yx_multi_number(_g(geometry, "maxh"), _g(geometry, "maxw"), r, n, int(x), C.byref(oy), C.byref(ox))
    if rc != 0:
        raise AssertionError("x2yxMultiNumber: class id %s not coherent with geometry" % x)  # :131 assert
    return int(oy.value), int(ox.value)


def x2yxMulti2(geometry, x, compat_c=False):
    """opticalflow_model_multiscale.lua:72-81 (+ x2yxMulti2.c): LongTensor of class ids -> rety, retx.
    compat_c=True reproduces the shipped C body bug for bug (see include/dfe.h)."""
    if x.dtype != torch.int64:
        raise TypeError("x2yxMulti2: LongTensor expected")  # luaT_checkudata(L, 1, idlong)
    x = x.contiguous()
    ctx = get_ctx(x)
    r, n = _ratios(geometry)
    rety = torch.zeros_like(x) if compat_c else torch.empty_like(x)
    retx = torch.zeros_like(x) if compat_c else torch.empty_like(x)
    ctx.check(
        lib().dfe_x2yx_multi(
            ctx.handle, _g(geometry, "maxh"), _g(geometry, "maxw"), r, n, ptr(x), x.numel(), ptr(rety), ptr(retx), 1 if compat_c else 0
        )
    )
    return rety, retx


def x2yxMulti(geometry, x):
    """opticalflow_model_multiscale.lua:54-70"""
    if isinstance(x, (int, float)):
        return x2yxMultiNumber(geometry, x)
    return x2yxMulti2(geometry, x)


def getMiddleIndex(geometry):
    """opticalflow_model.lua:36-43"""
    if _g(geometry, "multiscale"):
        return yx2xMulti(geometry, 0, 0)
    y, x = centered2onebased(geometry, 0, 0)
    return yx2x(geometry, y, x)


def rgb2y(rgb):
    """image.rgb2y: [3][H][W] -> [1][H][W] luminance (dfe_rgb2y_f32)."""
    if rgb.dim() != 3 or rgb.shape[0] != 3:
        raise ValueError("rgb2y: 3 x H x W expected")
    rgb = rgb.contiguous()
    y = torch.empty((1, rgb.shape[1], rgb.shape[2]), dtype=torch.float32, device=rgb.device)
    ctx = get_ctx(rgb)
    ctx.check(lib().dfe_rgb2y_f32(ctx.handle, ptr(rgb), rgb.shape[1], rgb.shape[2], ptr(y)))
    return y


def prepareInput(geometry, patch1, patch2, literal_rgb2y=False):
    """opticalflow_model.lua:131-151: what the drivers feed the model with.
      * prefilter: the patches must have the last layer's plane count (:133-134);
      * else, when the first layer reads one plane and the patches are RGB, they are turned into luminance (:136-138).  The script
        writes `patch1 = image.rgb2y(patch1, patch2)` -- image.rgb2y(src, dst) converts patch1 INTO patch2's storage and returns it, so
        read literally both inputs become the luminance of patch1 and the flow is zero everywhere.  Default here: each patch becomes its
        own luminance (what the model needs); literal_rgb2y=True reproduces the line as written;
      * multiscale: {patch1, patch2} as they are (the multiscale model pads and crops itself, :143-144);
      * single scale: patch1 narrowed to the model's output region -- rows ceil(maxh/2) .. + H - maxh + 1, columns ceil(maxw/2) ..
        (1-based, :147-149; the script's own TODO notes that the ground truth uses floor)."""
    if tuple(patch1.shape) != tuple(patch2.shape):
        raise AssertionError("prepareInput: patches of different sizes")   # assert(sameSize(patch1, patch2))
    layers = _g(geometry, "layers")
    if _g(geometry, "prefilter"):
        assert patch1.shape[0] == layers[-1][3]
    else:
        if layers[0][0] == 1 and patch1.shape[0] == 3:
            if literal_rgb2y:
                patch1 = patch2 = rgb2y(patch1)
            else:
                patch1, patch2 = rgb2y(patch1), rgb2y(patch2)
        assert patch1.shape[0] == layers[0][0]
    if _g(geometry, "multiscale"):
        return [patch1, patch2]
    maxh, maxw = _g(geometry, "maxh"), _g(geometry, "maxw")
    y0, x0 = math.ceil(maxh / 2) - 1, math.ceil(maxw / 2) - 1                  # narrow(2, ceil(maxh/2), H - maxh + 1), 1-based
    # (a view, as torch's narrow is: nn.SpatialMatching reads it in place -- dfe_spatial_matching_strided_f32)
    p1 = patch1[:, y0 : y0 + patch1.shape[1] - maxh + 1, x0 : x0 + patch1.shape[2] - maxw + 1]
    return [p1, patch2]


def getOutputConfidences(geometry, input, threshold=None):
    """opticalflow_model.lua:153-169.  input H x W x N (probabilities).  Without threshold: arg-max
    with the centre tie-break, confidences = 1.  With threshold: extractOutput(input, 0.11) and
    confidences = scores > threshold; the reference leaves imaxs/scores uninitialised for pixels
    with nothing above 0.11 -- here they are defined as imaxs = middleIndex, scores = 0."""
    if input.dim() != 3:
        raise ValueError("getOutputConfidences: input must be H x W x N")
    input = input.contiguous()
    H, W, N = input.shape
    middle = getMiddleIndex(geometry)
    ctx = get_ctx(input)
    if threshold is None:
        idx = torch.empty((H, W), dtype=torch.int64, device=input.device)
        ctx.check(lib().dfe_argbest_center(ctx.handle, ptr(input), H * W, N, middle, 1, ptr(idx), None))
        return idx, torch.ones((H, W), dtype=torch.float32, device=input.device)
    imaxs = torch.full((H, W), middle, dtype=torch.int64, device=input.device)
    scores = torch.zeros((H, W), dtype=torch.float32, device=input.device)
    extractoutput.extractOutput(input, scores, 0.11, imaxs)
    return imaxs, scores.gt(threshold)


def getOutputConfidences2(geometry, input):
    """opticalflow_model.lua:171-199 ('mean' extraction): soft arg-max y, x (1-based cell coordinates, float) and the
    confidence mask: extractOutput on the window's row marginal (threshold 0.11), confident where a score was written."""
    if input.dim() != 3:
        raise ValueError("getOutputConfidences2: input must be H x W x N")
    input = input.contiguous()
    H, W, N = input.shape
    maxh, maxw = _g(geometry, "maxh"), _g(geometry, "maxw")
    if N != maxh * maxw:
        raise ValueError("getOutputConfidences2: N != maxh*maxw")
    ctx = get_ctx(input)
    x = torch.empty((H, W), dtype=torch.float32, device=input.device)
    y = torch.empty_like(x)
    ctx.check(lib().dfe_output_extractor_f32(ctx.handle, ptr(input), H * W, maxh, maxw, ptr(x), ptr(y)))
    marg = torch.empty((H, W, maxh), dtype=torch.float32, device=input.device)
    ctx.check(lib().dfe_marginal_sum_f32(ctx.handle, ptr(input), H * W, maxh, maxw, ptr(marg)))
    imaxs = torch.zeros((H, W), dtype=torch.int64, device=input.device)
    scores = torch.zeros((H, W), dtype=torch.float32, device=input.device)   # uninitialised in the reference (:193)
    extractoutput.extractOutput(marg, scores, 0.11, imaxs)
    return y, x, scores.gt(0)


def processOutput(geometry, output, process_full=None, threshold=None):
    """opticalflow_model.lua:201-252: index, confidences, y, x and the centre-pasted full-frame flow (plane 0 = y,
    plane 1 = x), for output_extraction_method 'max' (arg-max with the centre tie-break / extractOutput) and 'mean'
    (soft arg-max, single scale only)."""
    ret = {}
    if _g(geometry, "output_extraction_method", "max") == "max":
        ret["index"], ret["confidences"] = getOutputConfidences(geometry, output, threshold)
        if _g(geometry, "multiscale"):
            ret["y"], ret["x"] = x2yxMulti(geometry, ret["index"])
        else:
            y, x = x2yx(geometry, ret["index"])
            yoff, xoff = centered2onebased(geometry, 0, 0)
            ret["y"], ret["x"] = y - yoff, x - xoff
    else:
        assert not _g(geometry, "multiscale")   # :219
        y, x, ret["confidences"] = getOutputConfidences2(geometry, output)
        ret["index"] = yx2x(geometry, torch.floor(y + 0.5), torch.floor(x + 0.5)).to(torch.int64)   # :221
        yoff, xoff = centered2onebased(geometry, 0, 0)
        ret["y"], ret["x"] = y - yoff, x - xoff
    if process_full is None:
        process_full = True
    if process_full:
        hImg, wImg = _g(geometry, "hImg"), _g(geometry, "wImg")
        h, w = ret["y"].shape
        ho, wo = (hImg - h) // 2, (wImg - w) // 2
        full = torch.zeros((2, hImg, wImg), dtype=torch.float32, device=output.device)
        full[0, ho : ho + h, wo : wo + w] = ret["y"].to(torch.float32)
        full[1, ho : ho + h, wo : wo + w] = ret["x"].to(torch.float32)
        ret["full"] = full
        fc = torch.zeros((hImg, wImg), dtype=torch.float32, device=output.device)
        fc[ho : ho + h, wo : wo + w] = ret["confidences"].to(torch.float32)
        ret["full_confidences"] = fc
    return ret
