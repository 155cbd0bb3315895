"""Radial / polar helpers with the reference's names (radial/cartesian2polar.lua,
radial/radial_opticalflow_polar.lua, radial/radial_opticalflow_display.lua)."""
import math

import torch

from ._lib import lib
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
