"""CPU suite, next-row N4: the oracle's relative-pose restatement (sfm2.getEgoMotion2's role: radial/radial_opticalflow_data.lua:211-231)
on synthetic two-view geometry with a known answer.  sfm2 is un-vendored: parity with the reference is unpinned; what is pinned
here is the geometry (the planted pose comes back, the epipole is K T, the rotation-only warp is undone by removeEgoMotion)."""
import numpy as np
import pytest

from tests import oracle as orc


def two_views(n=600, seed=0, outliers=0.25, noise=0.15, W=640, H=480):
    """n scene points in front of both cameras; returns (p1, p2 pixel coordinates, K, R, T) with x2 = R x1 + t, T = t / |t|."""
    rng = np.random.default_rng(seed)
    K = np.array([[520.0, 0, 320.0], [0, 515.0, 238.0], [0, 0, 1]])
    a, b, c = 0.03, -0.02, 0.015
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    R = Rz @ Ry @ Rx
    t = np.array([0.08, -0.03, -0.35])                    # mostly forward motion: the scene moves towards the camera
    X1 = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(3, 9, n)], 1)
    X2 = X1 @ R.T + t
    p1 = X1 @ K.T
    p1 = p1[:, :2] / p1[:, 2:]
    p2 = X2 @ K.T
    p2 = p2[:, :2] / p2[:, 2:]
    keep = (p1[:, 0] > 0) & (p1[:, 0] < W) & (p1[:, 1] > 0) & (p1[:, 1] < H) & (p2[:, 0] > 0) & (p2[:, 0] < W) & (p2[:, 1] > 0) & (p2[:, 1] < H)
    p1, p2 = p1[keep], p2[keep]
    p2 = p2 + rng.normal(0, noise, p2.shape)
    nout = int(outliers * len(p1))
    p2[:nout] = np.stack([rng.uniform(0, W, nout), rng.uniform(0, H, nout)], 1)
    return p1.astype(np.float32), p2.astype(np.float32), K, R, t / np.linalg.norm(t), nout


def rot_angle(Ra, Rb):
    return np.degrees(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))


def test_oracle_pose_recovers_the_planted_motion():
    p1, p2, K, R, T, nout = two_views()
    rc, Re, Te, ni, F = orc.ego_motion_from_points(p1, p2, K, 1.0, 512, 7)
    assert rc == 0
    assert abs(np.linalg.det(Re) - 1) < 1e-9 and np.abs(Re @ Re.T - np.eye(3)).max() < 1e-9 and abs(np.linalg.norm(Te) - 1) < 1e-9
    assert rot_angle(R, Re) < 0.15                              # degrees
    assert np.degrees(np.arccos(np.clip(Te @ T, -1, 1))) < 2.0  # the direction of translation (its length is unobservable)
    n = len(p1)
    assert n - nout - 25 <= ni <= n - nout + 12                 # the consensus set = the non-outliers, give or take the noise tail
    # the fundamental matrix has rank 2, unit norm, and the true correspondences satisfy it
    assert abs(np.linalg.norm(F) - 1) < 1e-9 and abs(np.linalg.det(F)) < 1e-12
    h1 = np.concatenate([p1[nout:], np.ones((n - nout, 1), np.float32)], 1).astype(np.float64)
    h2 = np.concatenate([p2[nout:], np.ones((n - nout, 1), np.float32)], 1).astype(np.float64)
    l = h1 @ F.T
    d = np.abs((h2 * l).sum(1)) / np.hypot(l[:, 0], l[:, 1])
    assert np.median(d) < 0.3
    # the epipole in the current frame is K T (radial_opticalflow_data.lua:218-219)
    rc, e_est = orc.epipole(K, Te)
    rc2, e_true = orc.epipole(K, T)
    assert rc == 0 == rc2 and np.hypot(e_est[0] - e_true[0], e_est[1] - e_true[1]) < 20.0   # (2 degrees of T at f = 520 px)
    # deterministic for a seed; weights <= 0 exclude correspondences (here: every planted outlier -> all inliers)
    rc, Re2, Te2, ni2, _ = orc.ego_motion_from_points(p1, p2, K, 1.0, 512, 7)
    assert np.array_equal(Re, Re2) and np.array_equal(Te, Te2) and ni == ni2
    w = np.ones(n, np.float32)
    w[:nout] = 0
    rc, Re3, Te3, ni3, _ = orc.ego_motion_from_points(p1, p2, K, 1.0, 256, 3, weights=w)
    assert rc == 0 and rot_angle(R, Re3) < 0.15 and ni3 >= n - nout - 25
    # too few points / nothing consistent
    assert orc.ego_motion_from_points(p1[:5], p2[:5], K, 1.0, 16, 1)[0] != 0
    rng = np.random.default_rng(1)
    junk = rng.uniform(0, 400, (60, 2)).astype(np.float32)
    assert orc.ego_motion_from_points(junk, rng.uniform(0, 400, (60, 2)).astype(np.float32), K, 0.05, 64, 1)[0] != 0


def test_oracle_rotation_warp_and_undistortion_properties():
    """removeEgoMotion undoes a pure rotation (the reason the callers apply it before the polar warp); undistortImage with zero
    coefficients is the identity; the epipole of a sideways translation is at infinity."""
    rng = np.random.default_rng(0)
    H, W = 60, 80
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    smooth = np.stack([np.sin(xs / 9) + np.cos(ys / 7)] * 2).astype(np.float32)
    K = np.array([[90.0, 0, 40.0], [0, 88.0, 30.0], [0, 0, 1]])
    a = 0.04
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    w1, m1 = orc.remove_ego_motion(smooth, K, R)
    back, m2 = orc.remove_ego_motion(w1, K, R, inverse=True)
    ok = (m2 > 0) & (np.roll(m1, 0) > 0)
    ok[:, :8] = ok[:, -8:] = False
    ok[:6] = ok[-6:] = False
    assert ok.sum() > 1000 and np.abs(back - smooth)[:, ok].max() < 3e-2
    assert np.array_equal(orc.remove_ego_motion(smooth, K, np.eye(3))[0], smooth)
    assert np.abs(orc.undistort_image(smooth, K, np.zeros(5)) - smooth).max() < 1e-5
    und = orc.undistort_image(smooth, K, [-0.38, 0.21, 0.003, 0.0009, -0.07])
    assert np.abs(und - smooth)[:, 25:35, 35:45].max() < 0.02 and np.abs(und - smooth).max() > 0.05   # the centre barely moves, the corners do
    assert orc.epipole(K, [1.0, 0.0, 0.0])[0] != 0
    del rng
