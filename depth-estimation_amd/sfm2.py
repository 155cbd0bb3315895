"""The slice of the un-vendored `sfm2` package the hot path's callers use (next-row N4), with its names:
undistortImage, removeEgoMotion, getEgoMotion2 (relative pose R, T by parallel RANSAC on the device; the correspondences are
the caller's tracks or samples of the matcher's dense flow instead of sfm2's own OpenCV corner tracks), getEpipole (e2 = K T
scaled, radial/radial_opticalflow_data.lua:218-220) and -- not in the reference -- getFOEFromFlow (focus of expansion of a
dense flow field, an estimator of this library for the pure-translation case).
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


def getEgoMotion2(K, flow=None, confidences=None, pts1=None, pts2=None, weights=None, maxPoints=400, ransacMaxDist=1.0, iterations=512, seed=0):
    """sfm2.getEgoMotion2{im1, im2, K, maxPoints, pointsQuality, ransacMaxDist, pointsMinDistance} -> R, T, nFound, nInliers, fundmat
    (radial/radial_opticalflow_data.lua:211-217; getEgoMotion: depth_estimation_api.lua:141).  The reference hands over the two
    IMAGES and sfm2 tracks corners itself; here the correspondences are given: either `flow` (2 x H x W, plane 0 = y, 1 = x: the
    matcher's dense flow from frame 1 to frame 2, sampled on a regular grid of at most maxPoints points, `confidences` <= 0
    skipped) or `pts1` / `pts2` (N x 2 (x, y) pixel positions, `weights` <= 0 skipped).  Returns R (3 x 3 float64 tensor), T (3,
    |T| = 1), nFound, nInliers, fundmat (3 x 3) with x2 ~ R x1 + T: getEpipole(K, T) is the FOE in the current frame and
    removeEgoMotion(prev, K, R, inverse=True) takes the rotation out of the previous frame."""
    R, T, F = (C.c_double * 9)(), (C.c_double * 3)(), (C.c_double * 9)()
    nf, ni = C.c_int(), C.c_int()
    if flow is not None:
        flow = flow.contiguous()
        _, H, W = flow.shape
        conf = confidences.contiguous() if confidences is not None else None
        ctx = get_ctx(flow)
        ctx.check(lib().dfe_ego_motion_from_flow_f32(ctx.handle, ptr(flow[0]), ptr(flow[1]), ptr(conf) if conf is not None else None, H, W, _d(K, 9), int(maxPoints),
                                                     float(ransacMaxDist), int(iterations), int(seed), R, T, C.byref(nf), C.byref(ni), F))
    else:
        if pts1 is None or pts2 is None:
            raise ValueError("getEgoMotion2: give the dense flow or two point lists")
        pts1, pts2 = pts1.to(torch.float32).contiguous(), pts2.to(torch.float32).contiguous()
        if pts1.dim() != 2 or pts1.shape[1] != 2 or tuple(pts1.shape) != tuple(pts2.shape):
            raise ValueError("getEgoMotion2: pts1 / pts2 must both be N x 2, got %s / %s" % (tuple(pts1.shape), tuple(pts2.shape)))
        w = weights.to(torch.float32).contiguous() if weights is not None else None
        ctx = get_ctx(pts1)
        ctx.check(lib().dfe_ego_motion_from_points_f32(ctx.handle, ptr(pts1), ptr(pts2), ptr(w) if w is not None else None, pts1.shape[0], _d(K, 9),
                                                       float(ransacMaxDist), int(iterations), int(seed), R, T, C.byref(ni), F))
        nf.value = int((w > 0).sum()) if w is not None else pts1.shape[0]
    return (torch.tensor(R[:], dtype=torch.float64).reshape(3, 3), torch.tensor(T[:], dtype=torch.float64), nf.value, ni.value,
            torch.tensor(F[:], dtype=torch.float64).reshape(3, 3))
