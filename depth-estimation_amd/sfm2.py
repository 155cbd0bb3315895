"""The slice of the un-vendored `sfm2` package the hot path's callers use (next-row N4), with its names:
undistortImage, removeEgoMotion, and -- in place of the sparse LK + RANSAC getEgoMotion2 -- getEpipole (e2 = K T scaled,
radial/radial_opticalflow_data.lua:218-220) and getFOEFromFlow (focus of expansion of a dense flow field).
Restated from the call sites (radial/radial_opticalflow_data.lua:24,211-231, depth_estimation_api.lua:139-147,
test_opticalflow.lua:280-284); `sfm2` itself is not in the reference repository, so parity is unpinned."""
import ctypes as C

import torch

from ._lib import lib
from .context import get_ctx, ptr


def _d(vals, n):
    v = [float(x) for x in (vals.reshape(-1).tolist() if hasattr(vals, "reshape") else vals)]
    if len(v) != n:
        raise ValueError("expected %d numbers, got %d" % (n, len(v)))
    return (C.c_double * n)(*v)


def undistortImage(img, K, distP):
    """sfm2.undistortImage(img, K, distP): img C x H x W FloatTensor, K 3 x 3, distP = (k1, k2, p1, p2, k3)."""
    img = img.contiguous()
    Cc, H, W = img.shape
    out = torch.empty_like(img)
    ctx = get_ctx(img)
    ctx.check(lib().dfe_undistort_image_f32(ctx.handle, ptr(img), Cc, H, W, _d(K, 9), _d(distP, 5), ptr(out)))
    return out


def removeEgoMotion(img, K, R, mode="bilinear", inverse=False):
    """sfm2.removeEgoMotion(img, K, R, 'bilinear') -> warped, mask: the rotation R between the two frames is undone by the
    homography K R K^-1 (inverse=True: R^T); mask is 1 where the warped pixel has a source inside the frame."""
    if mode != "bilinear":
        raise NotImplementedError("removeEgoMotion: the reference uses 'bilinear'")
    img = img.contiguous()
    Cc, H, W = img.shape
    out = torch.empty_like(img)
    mask = torch.empty((H, W), dtype=torch.float32, device=img.device)
    ctx = get_ctx(img)
    ctx.check(lib().dfe_remove_ego_motion_f32(ctx.handle, ptr(img), Cc, H, W, _d(K, 9), _d(R, 9), int(inverse), ptr(out), ptr(mask)))
    return out, mask


def getEpipole(K, T, scale=1.0):
    """e2 = K * T; e2 = e2 / e2[3]; e2 = e2 * scale (radial/radial_opticalflow_data.lua:218-220) -> (x, y)"""
    e = (C.c_double * 2)()
    rc = lib().dfe_epipole(_d(K, 9), _d(T, 3), float(scale), e)
    if rc != 0:
        raise ValueError("getEpipole: translation parallel to the image plane (epipole at infinity)")
    return e[0], e[1]


def getFOEFromFlow(flow, confidences=None, min_flow=0.5, iterations=2):
    """Focus of expansion (x, y) of a dense flow field 2 x H x W (plane 0 = y, plane 1 = x, the layout of processOutput's
    `full`): the point closest to all flow lines, Huber re-weighted `iterations` times.  Also returns the weight sum."""
    flow = flow.contiguous()
    _, H, W = flow.shape
    conf = confidences.contiguous() if confidences is not None else None
    out, n = (C.c_double * 2)(), C.c_double()
    ctx = get_ctx(flow)
    ctx.check(lib().dfe_foe_from_flow_f32(ctx.handle, ptr(flow[0]), ptr(flow[1]), ptr(conf) if conf is not None else None, H, W, float(min_flow),
                                          int(iterations), out, C.byref(n)))
    return (out[0], out[1]), n.value
