"""Radial / polar helpers with the reference's names (radial/cartesian2polar.lua,
radial/radial_opticalflow_polar.lua, radial/radial_opticalflow_display.lua)."""
import ctypes as C
import math

import torch

from ._lib import lib, RadialParams
from .context import get_ctx, ptr


def getRMax(h, w, e2):
    """radial/radial_opticalflow_polar.lua:4-10"""
    x, y = float(e2[0]), float(e2[1])
    return math.floor(math.sqrt(max(max(x * x + y * y, (w - x) * (w - x) + y * y),
                                    max(x * x + (h - y) * (h - y), (w - x) * (w - x) + (h - y) * (h - y)))))


def getC2PMask(wsrc, hsrc, wdst, hdst, xcenter=None, ycenter=None, lpadding=0, rpadding=0, rmax=None, alpha=None, device="cuda"):
    """radial/cartesian2polar.lua:4-49 -> FloatTensor 2 x hdst x (wdst+lpadding+rpadding)"""
    lpadding, rpadding = lpadding or 0, rpadding or 0
    if rmax is None:
        rmax = min(hsrc // 2, wsrc // 2) - 1
    xcenter = wsrc / 2 if xcenter is None else xcenter
    ycenter = hsrc / 2 if ycenter is None else ycenter
    alpha = 1.0 if alpha is None else alpha
    mask = torch.empty((2, hdst, wdst + lpadding + rpadding), dtype=torch.float32, device=device)
    ctx = get_ctx(mask)
    ctx.check(lib().dfe_polar_grid_c2p_f32(ctx.handle, wsrc, hsrc, wdst, hdst, xcenter, ycenter, lpadding, rpadding, rmax, alpha, ptr(mask)))
    return mask


def getP2CMask(wsrc, hsrc, wdst, hdst, xcenter=None, ycenter=None, rmax=None, alpha=None, device="cuda"):
    """radial/cartesian2polar.lua:51-89 -> FloatTensor 2 x hdst x wdst"""
    if rmax is None:
        rmax = min(hdst // 2, wdst // 2) - 1
    xcenter = wdst / 2 if xcenter is None else xcenter
    ycenter = hdst / 2 if ycenter is None else ycenter
    alpha = 1.0 if alpha is None else alpha
    mask = torch.empty((2, hdst, wdst), dtype=torch.float32, device=device)
    ctx = get_ctx(mask)
    ctx.check(lib().dfe_polar_grid_p2c_f32(ctx.handle, wsrc, hsrc, wdst, hdst, xcenter, ycenter, rmax, alpha, ptr(mask)))
    return mask


def cartesian2polar(img, mask):
    """radial/cartesian2polar.lua:91-93: image.warp(img, mask, 'bilinear', false)"""
    squeeze = img.dim() == 2
    if squeeze:
        img = img.unsqueeze(0)
    img, mask = img.contiguous(), mask.contiguous()
    Cc, H, W = img.shape
    _, Hd, Wd = mask.shape
    out = torch.empty((Cc, Hd, Wd), dtype=torch.float32, device=img.device)
    ctx = get_ctx(img)
    ctx.check(lib().dfe_warp_bilinear_f32(ctx.handle, ptr(img), Cc, H, W, ptr(mask), Hd, Wd, ptr(out)))
    return out[0] if squeeze else out


def flow2depth(networkp, flow, center=None, kinfty=0.65):
    """radial/radial_opticalflow_display.lua:6-58 -> (ret/infty, confs)"""
    H, W = flow.shape
    if center is None:
        center = (W / 2, H / 2)
    infty = getRMax(networkp["hImg"], networkp["wImg"], center) * kinfty
    flow = flow.contiguous()
    depth = torch.empty_like(flow)
    conf = torch.empty_like(flow)
    ctx = get_ctx(flow)
    ctx.check(lib().dfe_flow_to_depth_radial(ctx.handle, ptr(flow), H, W, float(center[0]), float(center[1]), infty, ptr(depth), ptr(conf)))
    return depth, conf


def getKOutput(networkp):
    """radial/radial_opticalflow_polar.lua:12-16"""
    hPolar = networkp["hInput"] - math.floor((networkp["hKernel"] - 1) / 2) - networkp["hWin"] + 1
    return hPolar / networkp["hInput"]


def getP2CMaskOF(networkp, e2, alpha_polar=None, device="cuda"):
    """radial/radial_opticalflow_polar.lua:18-30: the polar->cartesian grid for the (smaller) optical-flow output.
    wOutput / hOutput are not integers in general; Torch truncates tensor sizes, so does this."""
    wPolar = networkp["wInput"]
    hPolar = networkp["hInput"] - networkp["hKernel"] - networkp["hWin"] + 2
    kOutput = hPolar / networkp["hInput"]
    wOutput = networkp["wImg"] * kOutput
    hOutput = networkp["hImg"] * kOutput
    newRMax = getRMax(networkp["hImg"], networkp["wImg"], e2) * kOutput
    return getP2CMask(wPolar, hPolar, int(wOutput), int(hOutput), float(e2[0]) * kOutput, float(e2[1]) * kOutput, newRMax,
                      alpha_polar, device=device)


def computeDepthMapFromFlow(xflow, mask, imu_tx):
    """ARdroneAPI::computeDepthMapFromFlow (ardrone/ardrone_api.cpp:99-140) -> (depthMap, confidenceMap)"""
    xflow, mask = xflow.contiguous(), mask.contiguous()
    H, W = xflow.shape
    depth, conf = torch.empty_like(xflow), torch.empty_like(xflow)
    ctx = get_ctx(xflow)
    ctx.check(lib().dfe_flow_to_depth_ardrone(ctx.handle, ptr(xflow), ptr(mask), H, W, float(imu_tx), ptr(depth), ptr(conf)))
    return depth, conf


# ---- the radial network and the composed path (radial/radial_opticalflow_network.lua, radial/test_radial_opticalflow.lua) ----
def networkp_kernel_size(networkp):
    """hKernel / wKernel as the scripts derive them from the layer list (radial/train_radial_opticalflow.lua:89-96)."""
    hK = wK = 1
    for layer in networkp["layers"]:
        if isinstance(layer, (list, tuple)):
            hK += layer[1] - 1
            wK += layer[2] - 1
    return hK, wK


def getMatcher(networkp):
    """radial/radial_opticalflow_network.lua:32-34"""
    from .nn import SpatialRadialMatching

    return SpatialRadialMatching(networkp["hWin"])


class SpatialPadding:
    """nn.SpatialPadding(pad_l, pad_r, pad_t, pad_b) with the negative pads (= crops) the radial network uses
    (SpatialPadding(0, 0, 0, -hWin+1), radial_opticalflow_network.lua:59); positive pads zero-fill."""

    def __init__(self, pad_l, pad_r, pad_t, pad_b):
        self.pads = (int(pad_l), int(pad_r), int(pad_t), int(pad_b))
        self.output = None

    def forward(self, x):
        l, r, t, b = self.pads
        Cc, H, W = x.shape
        if min(l, r, t, b) >= 0 and max(l, r, t, b) > 0:
            out = torch.zeros((Cc, H + t + b, W + l + r), dtype=x.dtype, device=x.device)
            out[:, t : t + H, l : l + W] = x
        else:
            if max(l, r, t, b) > 0:
                raise NotImplementedError("SpatialPadding: mixed crop and pad")
            out = x[:, -t : H + b, -l : W + r].contiguous()
        self.output = out
        return out

    updateOutput = forward

    def backward(self, x, gradOutput, scale=1.0):
        l, r, t, b = self.pads
        if min(l, r, t, b) >= 0:
            Cc, H, W = gradOutput.shape
            self.gradInput = gradOutput[:, t : H - b, l : W - r].contiguous()
        else:
            gi = torch.zeros_like(x)
            Cc, H, W = x.shape
            gi[:, -t : H + b, -l : W + r] = gradOutput
            self.gradInput = gi
        return self.gradInput

    updateGradInput = lambda self, x, g: self.backward(x, g)

    def _all_modules(self):
        return []


class LogSoftMaxRows:
    """nn.Reshape(1, hWin) -> nn.Minus -> nn.LogSoftMax -> nn.Reshape(hWin) of getTrainerNetwork
    (radial/radial_opticalflow_network.lua:48-52) on a ... x hWin volume: log soft-MIN over the last dimension."""

    def __init__(self):
        self.output = self.gradInput = None

    def forward(self, x):
        x = x.contiguous()
        N = x.shape[-1]
        neg = (-x).contiguous()
        out = torch.empty_like(x)
        ctx = get_ctx(x)
        ctx.check(lib().dfe_log_softmax_f32(ctx.handle, ptr(neg), x.numel() // N, N, ptr(out)))
        self.output = out
        return out

    updateOutput = forward

    def backward(self, x, gradOutput, scale=1.0):
        go = gradOutput.contiguous()
        N = go.shape[-1]
        gi = torch.empty_like(go)
        ctx = get_ctx(go)
        ctx.check(lib().dfe_log_softmax_backward_f32(ctx.handle, ptr(self.output), ptr(go), go.numel() // N, N, ptr(gi)))
        self.gradInput = -gi            # through the Minus
        return self.gradInput

    updateGradInput = lambda self, x, g: self.backward(x, g)

    def _all_modules(self):
        return []


def getTrainerNetwork(networkp, device="cuda", generator=None):
    """radial/radial_opticalflow_network.lua:36-54: the tester network followed by Reshape(1, hWin), Minus, LogSoftMax,
    Reshape(hWin) -- log-probabilities over the hWin radial displacements (ClassNLLCriterion on top, train_radial:233-236)."""
    network = getTesterNetwork(networkp, device=device, generator=generator)
    network.add(LogSoftMaxRows())
    return network


def getTesterNetwork(networkp, device="cuda", generator=None):
    """radial/radial_opticalflow_network.lua:56-74: ParallelTable{ Sequential{SpatialPadding(0,0,0,-hWin+1), filter},
    filter:clone(shared weights) } -> SpatialRadialMatching(hWin).  `network.modules` keeps the reference's layout
    (getWeights reads network.modules[1].modules[2].modules, :76-90)."""
    from .network import Sequential, ParallelTable, getFilterRadial, _SharedFilter

    network = Sequential()
    filters = ParallelTable()
    seq_prev = Sequential()
    seq_prev.add(SpatialPadding(0, 0, 0, -networkp["hWin"] + 1))
    filt = getFilterRadial(networkp, device=device, generator=generator)
    seq_prev.add(filt)
    filters.add(seq_prev)
    filters.add(_SharedFilter(filt))
    network.add(filters)
    network.add(getMatcher(networkp))
    return network


def _separable_weights(network, networkp):
    """(w1, b1, w2, b2, tanh_between) of a tester network whose filter is conv(1 x kW) [tanh] conv(kH x 1), else None."""
    from .network import SpatialConvolution, Tanh

    mods = network.modules[0].modules[1].modules
    convs = [m for m in mods if isinstance(m, SpatialConvolution)]
    tanhs = [i for i, m in enumerate(mods) if isinstance(m, Tanh)]
    if len(convs) != 2 or len(mods) - len(tanhs) != 2 or tanhs not in ([], [1]):
        return None
    c1, c2 = convs
    if c1.kH != 1 or c2.kW != 1 or c2.nInputPlane != c1.nOutputPlane:
        return None
    return c1.weight, c1.bias, c2.weight, c2.bias, bool(tanhs)


def radial_out_shape(networkp):
    """(rows of the matcher output, hOutput, wOutput): getP2CMaskOF radial/radial_opticalflow_polar.lua:18-30"""
    hK, _ = networkp_kernel_size(networkp)
    hPolar = networkp["hInput"] - hK - networkp["hWin"] + 2
    k = hPolar / networkp["hInput"]
    return hPolar, int(networkp["hImg"] * k), int(networkp["wImg"] * k)


def radialFlowDepth(networkp, network, prev_img, img, e2, kinfty=0.65, alpha_polar=1.0, one_call=True, want_volume=False, zero_last_row=False):
    """radial/test_radial_opticalflow.lua:186-225 for one frame pair: polar warps of both frames around the epipole e2,
    getTesterNetwork:forward, min(3) - 1, back to cartesian through getP2CMaskOF, flow2depth.  prev_img is the previous
    frame after the caller's ego-motion correction (sfm2.removeEgoMotion: `sfm2.removeEgoMotion` of this package, or the caller's own).  Returns a dict
    polar_flow, flow (cartesian), depth, confs [, output = the matcher volume].
    one_call: everything inside dfe_radial_flow_depth_pair_f32 (default separable filter stacks); else the staged module
    calls -- same numbers bit for bit.
    zero_last_row: also zero the last polar flow row, as the trainer's display path does (train_radial:178-180:
    `test:sub(h,h,1,w):zero()`); test_radial:204-207 -- the default here -- does not."""
    hK, wK = networkp_kernel_size(networkp)
    prev_img, img = prev_img.contiguous(), img.contiguous()
    Cc, hImg, wImg = img.shape
    if (hImg, wImg) != (networkp["hImg"], networkp["wImg"]) or tuple(prev_img.shape) != tuple(img.shape):
        raise ValueError("radialFlowDepth: frames must be %dx%d (networkp.hImg x wImg), got %s / %s" % (networkp["hImg"], networkp["wImg"], tuple(prev_img.shape), tuple(img.shape)))
    hm, hOut, wOut = radial_out_shape(networkp)
    if hm < 1:
        raise ValueError("radialFlowDepth: hInput %d too small for kernel %d + window %d" % (networkp["hInput"], hK, networkp["hWin"]))
    sep = _separable_weights(network, networkp) if one_call else None
    dev = img.device
    ret = {}
    if sep is not None and networkp["hWin"] in (8, 12, 15, 16):
        w1, b1, w2, b2, th = sep
        prm = RadialParams(Cc, hImg, wImg, networkp["hInput"], networkp["wInput"], networkp["hWin"], w1.shape[0], w1.shape[3], w2.shape[0], w2.shape[2],
                           int(th), float(alpha_polar), float(kinfty), 1 if zero_last_row else 0)
        vol = torch.empty((hm, networkp["wInput"], networkp["hWin"]), dtype=torch.float32, device=dev) if want_volume else None
        pf = torch.empty((hm, networkp["wInput"]), dtype=torch.float32, device=dev)
        cart, depth, conf = (torch.empty((hOut, wOut), dtype=torch.float32, device=dev) for _ in range(3))
        ctx = get_ctx(img)
        ctx.check(lib().dfe_radial_flow_depth_pair_f32(ctx.handle, C.byref(prm), ptr(prev_img), ptr(img), float(e2[0]), float(e2[1]), ptr(w1), ptr(b1),
                                                       ptr(w2), ptr(b2), ptr(vol) if want_volume else None, ptr(pf), ptr(cart), ptr(depth), ptr(conf)))
        ret.update(polar_flow=pf, flow=cart, depth=depth, confs=conf)
        if want_volume:
            ret["output"] = vol
        return ret
    rmax = getRMax(hImg, wImg, e2)
    mask = getC2PMask(wImg, hImg, networkp["wInput"], networkp["hInput"], e2[0], e2[1], (wK - 1) // 2, -(-(wK - 1) // 2), rmax, alpha_polar, device=dev)
    polar_prev, polar_img = cartesian2polar(prev_img, mask), cartesian2polar(img, mask)
    output = network.forward([polar_prev, polar_img])
    H1, Wi, hW = output.shape
    imin = torch.empty((H1, Wi), dtype=torch.int64, device=dev)
    ctx = get_ctx(output)
    ctx.check(lib().dfe_argbest_center(ctx.handle, ptr(output), H1 * Wi, hW, 0, 0, ptr(imin), None))   # output:min(3): first minimum, no centre rule
    idx = (imin - 1).to(torch.float32)                   # idx:add(-1), test_radial:207
    if zero_last_row:
        idx[-1].zero_()                                  # train_radial:178-180 (the trainer's display path only)
    np2 = dict(networkp, hKernel=hK, wKernel=wK)
    p2c = getP2CMaskOF(np2, e2, alpha_polar, device=dev)
    cart = cartesian2polar(idx, p2c)
    kout = getKOutput(np2)
    depth, conf = flow2depth(np2, cart, (e2[0] * kout, e2[1] * kout), kinfty)
    ret.update(polar_flow=idx, flow=cart, depth=depth, confs=conf)
    if want_volume:
        ret["output"] = output
    return ret
