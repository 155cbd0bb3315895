"""The dense single-scale path as the reference states it most clearly
(radial/radial_opticalflow_groundtruth.lua:9-112 = version2/groundtruth.lua:7-110)."""
import math

import torch
import torch.nn.functional as F

from ._lib import lib
from .context import get_ctx, ptr


def unfold(img, wKer, hKer):
    """radial/radial_opticalflow_groundtruth.lua:9-21 (im2col; a view op, kept for API parity --
    the cost-volume op never materialises it)."""
    C, H, W = img.shape
    u = img.unfold(1, hKer, 1).unfold(2, wKer, 1)  # C x h x w x hKer x wKer
    h, w = u.shape[1], u.shape[2]
    return u.permute(0, 3, 4, 1, 2).reshape(C * hKer * wKer, h, w).contiguous()


def cross_correlation_pad_output(output, wWin, hWin, wKer, hKer):
    """radial/radial_opticalflow_groundtruth.lua:23-35"""
    l = (wWin - 1) // 2 + (wKer - 1) // 2
    r = math.ceil((wWin - 1) / 2) + math.ceil((wKer - 1) / 2)
    t = (hWin - 1) // 2 + (hKer - 1) // 2
    b = math.ceil((hWin - 1) / 2) + math.ceil((hKer - 1) / 2)
    return F.pad(output, (l, r, t, b))


def _adapt_mask(hWin, wWin, hKer, wKer, mask):
    """radial/radial_opticalflow_groundtruth.lua:37-63"""
    h, w = mask.shape
    new = torch.zeros_like(mask)
    ls = (wWin - 1) // 2 + (wKer - 1) // 2
    if ls > 0:
        new[:, ls:] += mask[:, : w - ls]
    rs = math.ceil((wWin - 1) / 2) + math.ceil((wKer - 1) / 2)
    if rs > 0:
        new[:, : w - rs] += mask[:, rs:]
    ts = (hWin - 1) // 2 + (hKer - 1) // 2
    if ts > 0:
        new[ts:, :] += mask[: h - ts, :]
    bs = math.ceil((hWin - 1) / 2) + math.ceil((hKer - 1) / 2)
    if bs > 0:
        new[: h - bs, :] += mask[bs:, :]
    return new.gt(3.9).to(mask.dtype)


def compute_cartesian_groundtruth_cross_correlation(groundtruthp, img1, img2, mask=None):
    """radial/radial_opticalflow_groundtruth.lua:66-112.  Returns flowp 4 x H x W:
    plane 0 = y flow, 1 = x flow, 2 = mask, 3 = extractOutput(cost, 0.21) scores.
    One fused libdfe call (dfe_ssd_flow_f32) replaces unfold + pad + SpatialMatching + min +
    tie-break + decode + extractOutput.  Pixels where nothing exceeds 0.21 keep scores = 0
    (the reference leaves them uninitialised)."""
    assert groundtruthp["type"] == "cross-correlation"
    p = groundtruthp["params"]
    hWin, wWin = p["hWin"], p["wWin"]
    hKer = p.get("hKernel", p.get("hKer"))
    wKer = p.get("wKernel", p.get("wKer"))
    img1 = img1.contiguous()
    img2 = img2.contiguous()
    Cc, H, W = img1.shape
    if mask is None:
        mask = torch.ones((H, W), dtype=torch.float32, device=img1.device)
    mask = _adapt_mask(hWin, wWin, hKer, wKer, mask)
    Ho, Wo = H - hKer + 1 - hWin + 1, W - wKer + 1 - wWin + 1
    ctx = get_ctx(img1)
    dev = img1.device
    idx = torch.empty((Ho, Wo), dtype=torch.int64, device=dev)
    fy = torch.empty((Ho, Wo), dtype=torch.float32, device=dev)
    fx = torch.empty((Ho, Wo), dtype=torch.float32, device=dev)
    scores = torch.zeros((Ho, Wo), dtype=torch.float32, device=dev)
    imaxs = torch.zeros((Ho, Wo), dtype=torch.int64, device=dev)
    ctx.check(
        lib().dfe_ssd_flow_f32(
            ctx.handle, ptr(img1), ptr(img2), Cc, H, W, hKer, wKer, hWin, wWin, 0.21,
            ptr(idx), None, ptr(fy), ptr(fx), ptr(scores), ptr(imaxs),
        )
    )
    flow = torch.stack([fy, fx, torch.ones_like(fy), scores])
    flowp = cross_correlation_pad_output(flow, wWin, hWin, wKer, hKer)
    flowp[2] *= mask
    return flowp
