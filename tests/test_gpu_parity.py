"""GPU suite (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs, against the committed golden fixtures, and -- at BASELINE.json's full sizes --
through size-independent properties.

Bars: integer-valued frames (uint8 values held in fp32) and every integer/index output are
BIT-EXACT.  Float frames: the reference-order kernel is bit-exact too; the tiled kernel sums in a
different order, so its cost volume is held to |gpu-cpu| <= 1e-5*|cpu| + 1e-6*max|cpu| (COST_RTOL /
COST_ATOL_FRAC below) and its arg-min may differ only where the two best costs are within that band."""
import math
import os

from ctypes import byref, c_double as C_double, c_int as C_int

import numpy as np
import pytest
import torch

from tests import oracle as orc
from tests import refpath as rp

pytestmark = pytest.mark.gpu

COST_RTOL = 1e-5
COST_ATOL_FRAC = 1e-6
GOLD = os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz")


def T(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def cost_close(gpu, cpu):
    tol = COST_RTOL * np.abs(cpu) + COST_ATOL_FRAC * np.abs(cpu).max()
    return np.abs(gpu.astype(np.float64) - cpu) <= tol


def tie_aware_equal(idx_gpu, idx_cpu, cost_cpu):
    """indices equal, or the gpu's pick is within the tolerance band of the cpu's best cost"""
    vol = cost_cpu.reshape(idx_cpu.shape + (-1,))
    pick = np.take_along_axis(vol, (idx_gpu[..., None] - 1), -1)[..., 0]
    best = np.take_along_axis(vol, (idx_cpu[..., None] - 1), -1)[..., 0]
    ok = (idx_gpu == idx_cpu) | (np.abs(pick - best) <= 2 * (COST_RTOL * np.abs(best) + COST_ATOL_FRAC * np.abs(vol).max()))
    return ok


# ------------------------------------------------------------------ cost volume, tiled kernel
@pytest.mark.parametrize(
    "H,W,C,k,hWin,wWin",
    [
        (80, 100, 3, 7, 33, 33),   # the judged configuration's kernel/window on a small frame
        (67, 46, 3, 7, 33, 33),    # exactly one tile high (Ho=29) and one tile wide (Wo=8)
        (70, 60, 3, 7, 17, 17),    # config 1 window
        (64, 72, 3, 7, 16, 16),    # even window (tests/time_matching.lua uses 16x16)
        (60, 64, 3, 7, 8, 8),      # multiscale window: exactly one wave of displacements
        (75, 90, 1, 7, 17, 13),    # luminance frames, non-square window
        (50, 56, 3, 5, 17, 17),
        (40, 44, 3, 3, 9, 11),
    ],
)
def test_cost_volume_tiled_bit_exact_on_integer_frames(dfe, cuda, H, W, C, k, hWin, wWin):
    f0, f1, _, _ = rp.synth_pair(H, W, C=C, seed=H + W, max_flow=min(hWin, wWin) // 2 - 1)
    cpu = orc.ssd_cost_volume(f0, f1, k, k, hWin, wWin)
    ctx = dfe.get_ctx(0)
    ctx.set_cost_volume_kernel(2)
    try:
        out = dfe.nn.SSDCostVolume(hWin, wWin, k, k).forward([T(f0, cuda), T(f1, cuda)])
        assert ctx.last_kernel() == "ssd_cv_tiled_kernel"
    finally:
        ctx.set_cost_volume_kernel(0)
    gpu = out.cpu().numpy()
    assert gpu.shape == cpu.shape
    assert np.array_equal(gpu, cpu)


@pytest.mark.parametrize("tyq", [2, 3, 4, 5])
def test_cost_volume_every_tile_height_bit_exact(dfe, cuda, tyq):
    # every TYQ instantiation, on a frame whose last tile row/column is shifted (Ho, Wo not multiples)
    H, W = 83, 71
    f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=tyq, max_flow=7)
    cpu = orc.ssd_cost_volume(f0, f1, 7, 7, 17, 17)
    ctx = dfe.get_ctx(0)
    ctx.set_cost_volume_kernel(2)
    ctx.set_cost_volume_tile(tyq)
    try:
        out = torch.full(cpu.shape, -1.0, device=cuda)
        t0, t1 = T(f0, cuda), T(f1, cuda)
        ctx.check(dfe.lib().dfe_ssd_cost_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, 7, 7, 17, 17, out.data_ptr()))
        assert ctx.last_kernel() == "ssd_cv_tiled_kernel"
    finally:
        ctx.set_cost_volume_kernel(0)
        ctx.set_cost_volume_tile(0)
    assert np.array_equal(out.cpu().numpy(), cpu)


@pytest.mark.parametrize("nq", [0, 1, 3, 4, 5, 6, 106, 107, 142])
@pytest.mark.parametrize("H,W", [(80, 100), (75, 47), (131, 90), (230, 64)])
def test_cost_volume_rowimg_kernel_bit_exact(dfe, cuda, nq, H, W):
    # the row-image kernel (all 1089 cells of a tile row in one block, LDS row image, aligned copy-out), 33x33 window, as a
    # column sweep (nq = 1; (230, 64): several segments of a column) and with static tiles of many heights (0 = auto, 3..6 = 6n - 6
    # rows, 100 + ty = ty rows: the height is a run-time value, a multiple of 6), on frames whose last tile row / column are shifted and whose
    # runs start at every alignment mod 128 B
    f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=H + nq, max_flow=9)
    cpu = orc.ssd_cost_volume(f0, f1, 7, 7, 33, 33)
    ctx = dfe.get_ctx(0)
    ctx.set_cost_volume_kernel(3)
    ctx.set_cost_volume_tile(nq)
    try:
        out = torch.full(cpu.shape, -1.0, device=cuda)
        t0, t1 = T(f0, cuda), T(f1, cuda)
        ty = nq - 100 if nq >= 100 else 6 * nq - 6
        if nq > 1 and (cpu.shape[0] < ty or ty % 6):   # forced static tiles need one full tile of rows and sweep whole groups of 6 rows; auto and the column sweep (nq = 1) adapt
            with pytest.raises(dfe.DfeError):
                ctx.check(dfe.lib().dfe_ssd_cost_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, 7, 7, 33, 33, out.data_ptr()))
            return
        ctx.check(dfe.lib().dfe_ssd_cost_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, 7, 7, 33, 33, out.data_ptr()))
        assert ctx.last_kernel() == "ssd_cv_rowimg_kernel"
    finally:
        ctx.set_cost_volume_kernel(0)
        ctx.set_cost_volume_tile(0)
    assert np.array_equal(out.cpu().numpy(), cpu)


@pytest.mark.parametrize("nq", [0, 1, 5])
def test_cost_volume_rowimg_kernel_luminance_bit_exact(dfe, cuda, nq):
    # the row-image kernel's C = 1 instantiation (auto / column sweep / static 24-row tiles) against the oracle
    H, W = 131, 90
    f0, f1, _, _ = rp.synth_pair(H, W, C=1, seed=nq, max_flow=9)
    cpu = orc.ssd_cost_volume(f0, f1, 7, 7, 33, 33)
    ctx = dfe.get_ctx(0)
    ctx.set_cost_volume_kernel(3)
    ctx.set_cost_volume_tile(nq)
    try:
        out = torch.full(cpu.shape, -1.0, device=cuda)
        t0, t1 = T(f0, cuda), T(f1, cuda)
        ctx.check(dfe.lib().dfe_ssd_cost_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 1, H, W, 7, 7, 33, 33, out.data_ptr()))
        assert ctx.last_kernel() == "ssd_cv_rowimg_kernel"
    finally:
        ctx.set_cost_volume_kernel(0)
        ctx.set_cost_volume_tile(0)
    assert np.array_equal(out.cpu().numpy(), cpu)


@pytest.mark.parametrize("H,W,C", [(75, 47, 3), (131, 90, 3), (90, 100, 1)])
def test_float_frames_deterministic_and_within_tolerance(dfe, cuda, H, W, C):
    """Float frames: edge tiles are shifted inwards and overlap their neighbours, i.e. two blocks store the same pixels --
    every kernel gives a pixel the same association of adds whatever the tiling, so every mode has to return the same
    bits on every run, stay within tolerance of the oracle, and the fused pipeline's indices have to match the oracle's
    wherever the minimum is unique."""
    f0, f1, _, (cx, cy) = rp.synth_pair(H, W, C=C, seed=H + C, integer=False, max_flow=9)
    cpu = orc.ssd_cost_volume(f0, f1, 7, 7, 33, 33)
    ctx = dfe.get_ctx(0)
    t0, t1 = T(f0, cuda), T(f1, cuda)
    for mode, tile in ((0, 0), (3, 1), (3, 4), (2, 0)):
        ctx.set_cost_volume_kernel(mode)
        ctx.set_cost_volume_tile(tile)
        try:
            outs = []
            for _ in range(3):
                out = torch.full(cpu.shape, -1.0, device=cuda)
                ctx.check(dfe.lib().dfe_ssd_cost_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, 7, 7, 33, 33, out.data_ptr()))
                outs.append(out)
        finally:
            ctx.set_cost_volume_kernel(0)
            ctx.set_cost_volume_tile(0)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (mode, tile)
        assert cost_close(outs[0].cpu().numpy(), cpu).all(), (mode, tile)
    # fused pipeline, three runs: identical, and consistent with the unfused volume of the same kernel family
    res = []
    for _ in range(3):
        Ho, Wo = cpu.shape[:2]
        idx = torch.empty((Ho, Wo), dtype=torch.int64, device=cuda)
        best = torch.empty((Ho, Wo), dtype=torch.float32, device=cuda)
        ctx.check(dfe.lib().dfe_ssd_flow_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, 7, 7, 33, 33, 0.21,
                                            idx.data_ptr(), best.data_ptr(), None, None, None, None))
        res.append((idx, best))
    assert all(torch.equal(res[0][0], r[0]) and torch.equal(res[0][1], r[1]) for r in res[1:])
    mid = rp.middle_index(33, 33)
    idx_cpu, _ = orc.argbest_center(cpu.reshape(cpu.shape[0], cpu.shape[1], -1), mid, False)
    assert tie_aware_equal(res[0][0].cpu().numpy(), idx_cpu, cpu).all()


def test_cost_volume_tiled_float_frames_within_tolerance(dfe, cuda):
    f0, f1, _, _ = rp.synth_pair(80, 100, C=3, seed=2, integer=False)
    cpu = orc.ssd_cost_volume(f0, f1, 7, 7, 33, 33)
    ctx = dfe.get_ctx(0)
    ctx.set_cost_volume_kernel(2)
    try:
        gpu_t = dfe.nn.SSDCostVolume(33, 33, 7, 7).forward([T(f0, cuda), T(f1, cuda)])
    finally:
        ctx.set_cost_volume_kernel(0)
    gpu = gpu_t.cpu().numpy()
    assert cost_close(gpu, cpu).all()
    mid = rp.middle_index(33, 33)
    idx_cpu, _ = orc.argbest_center(cpu.reshape(cpu.shape[0], cpu.shape[1], -1), mid, False)
    idx = torch.empty(cpu.shape[:2], dtype=torch.int64, device=cuda)
    ctx.check(dfe.lib().dfe_argbest_center(ctx.handle, gpu_t.data_ptr(), idx.numel(), 33 * 33, mid, 0, idx.data_ptr(), None))
    ok = tie_aware_equal(idx.cpu().numpy(), idx_cpu, cpu)
    assert ok.all()
    # and the launch geometry must not matter: same result twice, bitwise
    ctx.set_cost_volume_kernel(2)
    try:
        again = dfe.nn.SSDCostVolume(33, 33, 7, 7).forward([T(f0, cuda), T(f1, cuda)]).cpu().numpy()
    finally:
        ctx.set_cost_volume_kernel(0)
    assert np.array_equal(gpu, again)


# ------------------------------------------------------------------ reference-order kernel: bit-exact on floats
@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_cost_volume_ref_kernel_bit_exact_on_floats(dfe, cuda, case):
    im1, im2, _, (hK, wK, hW, wW) = rp.kat_case(case)
    cpu = orc.ssd_cost_volume(im1, im2, hK, wK, hW, wW)
    ctx = dfe.get_ctx(0)
    ctx.set_cost_volume_kernel(1)
    try:
        gpu = dfe.nn.SSDCostVolume(hW, wW, hK, wK).forward([T(im1, cuda), T(im2, cuda)]).cpu().numpy()
        assert ctx.last_kernel() == "ssd_cv_ref_kernel"
    finally:
        ctx.set_cost_volume_kernel(0)
    assert np.array_equal(gpu, cpu)


def test_auto_dispatch_falls_back_when_frame_smaller_than_a_tile(dfe, cuda):
    f0, f1, _, _ = rp.synth_pair(19, 40, C=3, seed=1, max_flow=2)   # Ho = 19-6-8 = 5 < the smallest tile (6 rows)
    cpu = orc.ssd_cost_volume(f0, f1, 7, 7, 9, 9)
    ctx = dfe.get_ctx(0)
    gpu = dfe.nn.SSDCostVolume(9, 9, 7, 7).forward([T(f0, cuda), T(f1, cuda)]).cpu().numpy()
    assert ctx.last_kernel() == "ssd_cv_ref_kernel"
    assert np.array_equal(gpu, cpu)
    ctx.set_cost_volume_kernel(2)
    try:
        with pytest.raises(dfe.DfeError) as e:
            dfe.nn.SSDCostVolume(9, 9, 7, 7).forward([T(f0, cuda), T(f1, cuda)])
        assert e.value.code == -5
    finally:
        ctx.set_cost_volume_kernel(0)


# ------------------------------------------------------------------ nn.SpatialMatching / SpatialRadialMatching
def test_spatial_matching_module_bit_exact(dfe, cuda):
    rng = np.random.default_rng(0)
    # tests/time_matching.lua geometry scaled down: K=10 features, 16x16 window
    K, H1, W1, mh, mw = 10, 21, 33, 16, 16
    in1 = rng.standard_normal((K, H1, W1)).astype(np.float32)
    in2 = rng.standard_normal((K, H1 + mh - 1, W1 + mw - 1)).astype(np.float32)
    m = dfe.nn.SpatialMatching(mh, mw, False)
    out = m.forward([T(in1, cuda), T(in2, cuda)])
    assert tuple(out.shape) == (H1, W1, mh, mw) and m.output is out
    assert np.array_equal(out.cpu().numpy(), orc.spatial_matching(in1, in2, mh, mw))
    with pytest.raises(ValueError):
        m.forward([T(in1, cuda), T(in2[:, :-1], cuda)])
    with pytest.raises(TypeError):
        m.forward([T(in1, cuda).double(), T(in2, cuda).double()])


@pytest.mark.parametrize("K,H1,W1,mh,mw", [
    (3, 5, 253, 16, 16),     # the narrowest frame the flat-tile kernel takes (64 groups of 4 pixels per row), ragged last group
    (4, 7, 256, 16, 16),     # rows of exactly one tile
    (5, 9, 301, 17, 17),     # 17 window rows on 16 waves (row 16 = the extra task), tiles running over the row ends, W1 % 4 = 1
    (2, 3, 608, 17, 17),     # version2's width (640 - 32)
    (6, 11, 625, 16, 16),    # VGA minus the window: the verdict's shape, few rows
    (3, 6, 290, 8, 16),      # fewer window rows than waves: 8 x 16
    (3, 6, 327, 12, 17),     # 17 wide, 12 high (no extra task)
    (1, 1, 260, 16, 16),     # a single output row, one plane
    (3, 6, 270, 14, 16),     # half tiles of 7 window rows (fm_split: two blocks per tile), 112-float half windows
    (2, 4, 300, 10, 17),     # half tiles of 5 rows of a 17-wide window (unaligned half windows)
])
def test_spatial_matching_flat_kernel_bit_exact(dfe, cuda, K, H1, W1, mh, mw):
    """nn.SpatialMatching with 16- / 17-wide windows on wide feature maps (opticalflow_model.lua:93, version2/network.lua:30,
    tests/time_matching.lua:18): the flat-tile kernel (csrc/feat_matching_flat.hip) -- bit-exact against the oracle's k-ordered,
    separately rounded sum, equal to the round-3 kernels (fm_flat = 0) bit for bit, every output element written exactly once
    (the buffer is pre-filled with NaN)."""
    rng = np.random.default_rng(K * 1000 + W1 + mh)
    in1 = rng.standard_normal((K, H1, W1)).astype(np.float32)
    in2 = rng.standard_normal((K, H1 + mh - 1, W1 + mw - 1)).astype(np.float32)
    ctx = dfe.get_ctx(0)
    out = torch.full((H1, W1, mh, mw), float("nan"), device=cuda)
    t1, t2 = T(in1, cuda), T(in2, cuda)
    ctx.check(dfe.lib().dfe_spatial_matching_f32(ctx.handle, t1.data_ptr(), t2.data_ptr(), K, H1, W1, mh, mw, out.data_ptr()))
    torch.cuda.synchronize()
    assert ctx.last_kernel() == "feat_matching_flat_kernel", ctx.last_kernel()
    ref = orc.spatial_matching(in1, in2, mh, mw)
    got = out.cpu().numpy()
    assert not np.isnan(got).any(), "cells left unwritten: %d" % int(np.isnan(got).sum())
    assert np.array_equal(got, ref)
    with ctx.options(fm_flat=0):
        old = dfe.nn.SpatialMatching(mh, mw, False).forward([t1, t2])
        assert ctx.last_kernel() != "feat_matching_flat_kernel"
    assert torch.equal(old, out)
    for split in (0, 2, 4):                                          # one block per tile / two half blocks / four quarter blocks (16 rows only)
        with ctx.options(fm_split=split):
            other = dfe.nn.SpatialMatching(mh, mw, False).forward([t1, t2])
            assert ctx.last_kernel() == "feat_matching_flat_kernel"
        assert torch.equal(other, out), split


@pytest.mark.parametrize("mh,mw", [(17, 17), (16, 16), (10, 16)])
def test_spatial_matching_flat_kernel_output_is_bounded(dfe, cuda, mh, mw):
    """The flat-tile kernel writes nothing outside its output: guard floats in front of and behind an (unaligned) output buffer stay
    untouched, and an output that starts 4 bytes off a 16-B boundary (partial head / tail lines everywhere; with half tiles the
    general copy path instead of the two-windows-per-wave one) is still bit-exact."""
    rng = np.random.default_rng(77)
    K, H1, W1 = 3, 4, 270
    in1 = rng.standard_normal((K, H1, W1)).astype(np.float32)
    in2 = rng.standard_normal((K, H1 + mh - 1, W1 + mw - 1)).astype(np.float32)
    n = H1 * W1 * mh * mw
    buf = torch.full((n + 64 + 1,), -7.0, device=cuda)
    view = buf[33 : 33 + n]                                  # 4-B aligned only
    ctx = dfe.get_ctx(0)
    t1, t2 = T(in1, cuda), T(in2, cuda)
    ctx.check(dfe.lib().dfe_spatial_matching_f32(ctx.handle, t1.data_ptr(), t2.data_ptr(), K, H1, W1, mh, mw, view.data_ptr()))
    torch.cuda.synchronize()
    assert ctx.last_kernel() == "feat_matching_flat_kernel"
    h = buf.cpu().numpy()
    assert (h[:33] == -7.0).all() and (h[33 + n :] == -7.0).all()
    assert np.array_equal(h[33 : 33 + n].reshape(H1, W1, mh, mw), orc.spatial_matching(in1, in2, mh, mw))


def test_spatial_radial_matching_module_bit_exact(dfe, cuda):
    rng = np.random.default_rng(1)
    K, H1, W, hWin = 10, 40, 64, 15   # radial/train_radial_opticalflow.lua:27-29 defaults: 10 features, hWin=15
    in1 = rng.standard_normal((K, H1, W)).astype(np.float32)
    in2 = rng.standard_normal((K, H1 + hWin - 1, W)).astype(np.float32)
    out = dfe.nn.SpatialRadialMatching(hWin).forward([T(in1, cuda), T(in2, cuda)])
    cpu = orc.radial_matching(in1, in2, hWin)
    assert np.array_equal(out.cpu().numpy(), cpu)
    # consumer: test:min(3) - 1  (radial/train_radial_opticalflow.lua:166-167)
    assert np.array_equal(out.argmin(2).cpu().numpy(), cpu.argmin(2))


# ------------------------------------------------------------------ dense path end to end (KAT)
@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_kat_planted_flow_recovered_on_gpu(dfe, cuda, case):
    im1, im2, fb, (hK, wK, hW, wW) = rp.kat_case(case)
    gp = dict(type="cross-correlation", params=dict(hWin=hW, wWin=wW, hKer=hK, wKer=wK))
    flow = dfe.compute_cartesian_groundtruth_cross_correlation(gp, T(im1, cuda), T(im2, cuda)).cpu().numpy()
    assert flow.shape == (4,) + im1.shape[1:]
    assert flow[2].sum() > 0
    assert np.abs((fb - flow[:2]) * flow[2]).sum() == 0   # radial/radial_opticalflow_groundtruth.lua:186-192
    ref = rp.dense_flow_oracle(im1, im2, hW, wW, hK, wK)["flowp"]
    assert np.array_equal(flow[:3], ref[:3])
    assert np.allclose(flow[3], ref[3], rtol=1e-5, atol=1e-5)


def test_fused_flow_entry_bit_exact_on_integer_frames(dfe, cuda):
    f0, f1, flow, _ = rp.synth_pair(96, 128, C=3, seed=9, max_flow=7, noise_sigma=0)
    ref = rp.dense_flow_oracle(f0, f1, 17, 17, 7, 7)
    Ho, Wo = ref["idx"].shape
    ctx = dfe.get_ctx(0)
    idx = torch.empty((Ho, Wo), dtype=torch.int64, device=cuda)
    best = torch.empty((Ho, Wo), dtype=torch.float32, device=cuda)
    fy, fx = torch.empty_like(best), torch.empty_like(best)
    scores = torch.full((Ho, Wo), -2.0, device=cuda)
    imaxs = torch.full((Ho, Wo), -5, dtype=torch.int64, device=cuda)
    t0, t1 = T(f0, cuda), T(f1, cuda)   # keep the device frames alive across the asynchronous call
    ctx.check(dfe.lib().dfe_ssd_flow_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, 96, 128, 7, 7, 17, 17, 0.21,
                                        idx.data_ptr(), best.data_ptr(), fy.data_ptr(), fx.data_ptr(), scores.data_ptr(), imaxs.data_ptr()))
    assert np.array_equal(idx.cpu().numpy(), ref["idx"])
    assert np.array_equal(best.cpu().numpy(), ref["best"])
    assert np.array_equal(fy.cpu().numpy(), ref["fy"]) and np.array_equal(fx.cpu().numpy(), ref["fx"])
    sc, im = np.full((Ho, Wo), -2.0, np.float32), np.full((Ho, Wo), -5, np.int64)
    orc.extract_output(ref["cost"].reshape(Ho, Wo, -1), 0.21, im, sc)
    assert np.array_equal(imaxs.cpu().numpy(), im) and np.array_equal(scores.cpu().numpy(), sc)
    # noise-free planted flow is recovered wherever the source patch stayed inside the frame
    t = 8 + 3
    inner = (slice(t, 96 - t), slice(t, 128 - t))
    got_y = np.pad(ref["fy"], ((t, t), (t, t)))[inner]
    assert (np.abs(got_y - flow[0][inner]) <= 0).mean() > 0.9


@pytest.mark.parametrize(
    "hWin,wWin,C,thr",
    [
        (33, 33, 3, 0.21),      # 18 chunks, last one a single cell; M = 4
        (17, 17, 3, 0.11),      # 5 chunks, last one 33 cells; M = 8
        (16, 16, 1, 0.21),      # exactly 4 chunks, luminance
        (8, 8, 3, 0.21),        # one chunk: nothing beyond chunk 0 to look at
        (33, 33, 3, 20000.0),   # most pixels have fewer than 4 hits in their first 64 cells -> device-side fallback pass
        (17, 17, 3, 1e12),      # nothing above the threshold anywhere -> scores / imaxs untouched
    ],
)
@pytest.mark.parametrize("mode", [0, 2, 101, 104])
def test_fused_build_matches_build_plus_tail(dfe, cuda, hWin, wWin, C, thr, mode):
    """dfe_ssd_flow_f32 through the fused build (minima, centre cost and extractOutput leave the cost-volume kernel)
    == oracle, including the fallback for pixels whose first cells hold too few values above the threshold.  Auto mode
    takes the row-image kernel at 33x33 / C=3 and the tiled kernel elsewhere; mode 2 forces the tiled kernel; 101 / 104 =
    auto mode with the row-image kernel forced to its column sweep / to static 18-row tiles."""
    tile = 0
    if mode >= 100:
        if not (hWin == 33 and C == 3):
            pytest.skip("tile codes of the row-image kernel")
        mode, tile = 0, mode - 100
    H, W = 90, 110
    want = "ssd_cv_rowimg_kernel+fused_tail" if (mode == 0 and hWin == 33 and C == 3) else "ssd_cv_tiled_kernel+fused_tail"   # (+finalize: see the assert)
    f0, f1, _, _ = rp.synth_pair(H, W, C=C, seed=hWin + C, max_flow=min(hWin, wWin) // 2 - 1, noise_sigma=1.0)
    ref = rp.dense_flow_oracle(f0, f1, hWin, wWin, 7, 7, thr=thr)
    Ho, Wo = ref["idx"].shape
    ctx = dfe.get_ctx(0)
    # whole volume in one band, then ~2 MiB bands (auto mode only: a short last band needs the reference-order fallback)
    for limit in ((None, 2 << 20) if mode == 0 else (None,)):
        if limit:
            ctx.check(dfe.lib().dfe_set_scratch_limit(ctx.handle, limit))
        ctx.set_cost_volume_kernel(mode)
        ctx.set_cost_volume_tile(tile)
        try:
            idx = torch.empty((Ho, Wo), dtype=torch.int64, device=cuda)
            best = torch.empty((Ho, Wo), dtype=torch.float32, device=cuda)
            fy, fx = torch.empty_like(best), torch.empty_like(best)
            scores = torch.full((Ho, Wo), -2.0, device=cuda)
            imaxs = torch.full((Ho, Wo), -5, dtype=torch.int64, device=cuda)
            t0, t1 = T(f0, cuda), T(f1, cuda)
            ctx.check(dfe.lib().dfe_ssd_flow_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, 7, 7, hWin, wWin, thr,
                                                idx.data_ptr(), best.data_ptr(), fy.data_ptr(), fx.data_ptr(), scores.data_ptr(), imaxs.data_ptr()))
            if not limit:   # (a short last band may legitimately fall back to the reference-order kernel + tail pass)
                assert ctx.last_kernel().startswith(want)
        finally:
            ctx.set_cost_volume_kernel(0)
            ctx.set_cost_volume_tile(0)
            ctx.check(dfe.lib().dfe_set_scratch_limit(ctx.handle, 16 << 30))
        assert np.array_equal(idx.cpu().numpy(), ref["idx"])
        assert np.array_equal(best.cpu().numpy(), ref["best"])
        assert np.array_equal(fy.cpu().numpy(), ref["fy"]) and np.array_equal(fx.cpu().numpy(), ref["fx"])
        sc, im = np.full((Ho, Wo), -2.0, np.float32), np.full((Ho, Wo), -5, np.int64)
        orc.extract_output(ref["cost"].reshape(Ho, Wo, -1), thr, im, sc)
        assert np.array_equal(imaxs.cpu().numpy(), im) and np.array_equal(scores.cpu().numpy(), sc)


@pytest.mark.parametrize("delta", [0.0, -1e-9, 1e-9, -0.5])
def test_fused_extract_threshold_boundary(dfe, cuda, delta):
    """extractOutput's `value > threshold` compares a float with a double (extract_output.cpp:86).  Frames of a few small integer
    values make costs that EQUAL the threshold: thresholds at a frequent cost value, a hair below and above it (not representable in
    fp32) and half an integer below give the oracle's scores and indices."""
    H, W, hWin = 60, 84, 33
    f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=5, max_flow=6)
    f0, f1 = np.floor(f0 / 64).astype(np.float32), np.floor(f1 / 64).astype(np.float32)    # values 0 .. 3: many equal costs
    ref = rp.dense_flow_oracle(f0, f1, hWin, hWin, 7, 7, thr=0.21)
    Ho, Wo = ref["idx"].shape
    lead = ref["cost"].reshape(Ho, Wo, -1)[:, :, :8]
    vals, counts = np.unique(lead, return_counts=True)
    v = float(vals[np.argmax(counts)])                                                      # the most frequent cost among the lead cells
    thr = v + delta
    ctx = dfe.get_ctx(0)
    idx = torch.empty((Ho, Wo), dtype=torch.int64, device=cuda)
    best = torch.empty((Ho, Wo), dtype=torch.float32, device=cuda)
    fy, fx = torch.empty_like(best), torch.empty_like(best)
    scores = torch.full((Ho, Wo), -2.0, device=cuda)
    imaxs = torch.full((Ho, Wo), -5, dtype=torch.int64, device=cuda)
    t0, t1 = T(f0, cuda), T(f1, cuda)
    ctx.check(dfe.lib().dfe_ssd_flow_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, 7, 7, hWin, hWin, thr, idx.data_ptr(), best.data_ptr(),
                                        fy.data_ptr(), fx.data_ptr(), scores.data_ptr(), imaxs.data_ptr()))
    assert ctx.last_kernel().startswith("ssd_cv_rowimg_kernel+fused_tail")
    sc, im = np.full((Ho, Wo), -2.0, np.float32), np.full((Ho, Wo), -5, np.int64)
    orc.extract_output(ref["cost"].reshape(Ho, Wo, -1), thr, im, sc)
    assert (lead == np.float32(v)).sum() > 100                                             # the boundary is really exercised
    assert np.array_equal(imaxs.cpu().numpy(), im) and np.array_equal(scores.cpu().numpy(), sc)


# ------------------------------------------------------------------ per-pixel consumers
def test_argbest_center_device(dfe, cuda):
    rng = np.random.default_rng(5)
    for N, P in [(9, 7), (64, 300), (289, 129), (1089, 70), (160, 1000)]:
        vol = rng.integers(0, 6, size=(P, N)).astype(np.float32)   # many exact ties
        vol[0] = 3.0
        mid = N // 2 + 1
        ctx = dfe.get_ctx(0)
        for take_max in (0, 1):
            for middle in (mid, 0):
                idx = torch.empty(P, dtype=torch.int64, device=cuda)
                best = torch.empty(P, dtype=torch.float32, device=cuda)
                tv = T(vol, cuda)
                ctx.check(dfe.lib().dfe_argbest_center(ctx.handle, tv.data_ptr(), P, N, middle, take_max, idx.data_ptr(), best.data_ptr()))
                ei, eb = orc.argbest_center(vol, middle, take_max)
                assert np.array_equal(idx.cpu().numpy(), ei) and np.array_equal(best.cpu().numpy(), eb)


@pytest.mark.parametrize("N", [9, 64, 160, 289, 1089])
@pytest.mark.parametrize("thr", [0.11, 0.21, 0.0])
def test_extract_output_device(dfe, cuda, N, thr):
    rng = np.random.default_rng(N)
    H, W = 13, 17
    x = rng.random((H, W, N)).astype(np.float32)
    x = x / x.sum(-1, keepdims=True) * rng.choice([1.0, 4.0, 40.0], size=(H, W, 1)).astype(np.float32)
    x[0, 0] = 0.0                # nothing above threshold -> untouched
    x[0, 1, :5] = 0.5            # exact ties among the kept values
    x[0, 2, N - 1] = 0.9         # the hit is the very last cell
    scores = torch.full((H, W), -3.0, device=cuda)
    imaxs = torch.full((H, W), -7, dtype=torch.int64, device=cuda)
    dfe.extractoutput.extractOutput(T(x, cuda), scores, thr, imaxs)
    es, ei = np.full((H, W), -3.0, np.float32), np.full((H, W), -7, np.int64)
    orc.extract_output(x, thr, ei, es)
    assert np.array_equal(imaxs.cpu().numpy(), ei)
    assert np.array_equal(scores.cpu().numpy(), es)
    ret = torch.full((H, W), -7, dtype=torch.int64, device=cuda)
    gd = torch.full((H, W), 9, dtype=torch.int64, device=cuda)
    dfe.extractoutput.extractOutputMarginalized(T(x, cuda), thr, 0.8, ret, gd)
    er, eg = np.full((H, W), -7, np.int64), np.full((H, W), 9, np.int64)
    orc.extract_output_marginalized(x, thr, 0.8, er, eg)
    assert np.array_equal(ret.cpu().numpy(), er) and np.array_equal(gd.cpu().numpy(), eg)


def test_extractoutput_argument_checks(dfe, cuda):
    x = torch.zeros((4, 5, 9), device=cuda)
    with pytest.raises(TypeError):
        dfe.extractoutput.extractOutput(x.double(), torch.zeros((4, 5), device=cuda), 0.1, torch.zeros((4, 5), dtype=torch.int64, device=cuda))
    with pytest.raises(ValueError):
        dfe.extractoutput.extractOutput(x, torch.zeros((4, 4), device=cuda), 0.1, torch.zeros((4, 5), dtype=torch.int64, device=cuda))
    # empty input is a no-op, not an error
    dfe.extractoutput.extractOutput(torch.zeros((0, 5, 9), device=cuda), torch.zeros((0, 5), device=cuda), 0.1,
                                    torch.zeros((0, 5), dtype=torch.int64, device=cuda))


@pytest.mark.parametrize("maxh,maxw,ratios", [(8, 8, [1, 2, 4]), (4, 4, [1, 2, 4, 8]), (16, 16, [1, 2, 4, 8]), (8, 8, [1, 2]), (8, 8, [1])])
def test_x2yx_multi_device(dfe, cuda, maxh, maxw, ratios):
    geo = dict(maxh=maxh, maxw=maxw, ratios=ratios, multiscale=True)
    n = orc.multi_nclasses(maxh, maxw, ratios)
    rng = np.random.default_rng(n)
    ids = np.concatenate([np.arange(1, n + 1), rng.integers(1, n + 1, size=1000)]).astype(np.int64)
    ids = np.ascontiguousarray(ids[: ids.size // 8 * 8].reshape(-1, 8))
    rety, retx = dfe.x2yxMulti(geo, T(ids, cuda))
    _, ey, ex = orc.x2yx_multi(maxh, maxw, ratios, ids)
    assert np.array_equal(rety.cpu().numpy(), ey) and np.array_equal(retx.cpu().numpy(), ex)
    # round trip through the scalar encoder (tests/test_multiscale.lua:78-80)
    for i in (1, n // 2, n):
        y, x = dfe.x2yxMulti(geo, i)
        assert dfe.yx2xMulti(geo, y, x) == i
    bad = ids.copy()
    bad[0, 0] = n + 1
    with pytest.raises(dfe.DfeError):
        dfe.x2yxMulti(geo, T(bad, cuda))
    if len(ratios) > 1:
        cy, cx = dfe.x2yxMulti2(geo, T(ids, cuda), compat_c=True)
        oy, ox = orc.x2yx_multi_compat_c(maxh, maxw, ratios, ids, fill=0)
        assert np.array_equal(cy.cpu().numpy(), oy) and np.array_equal(cx.cpu().numpy(), ox)


def test_x2yx_and_process_output_single_scale(dfe, cuda):
    rng = np.random.default_rng(2)
    for maxh, maxw in [(17, 17), (16, 16), (12, 15)]:
        geo = dict(maxh=maxh, maxw=maxw, multiscale=False, hImg=40, wImg=50, output_extraction_method="max")
        H, W = 40 - maxh - 6 + 2, 50 - maxw - 6 + 2
        prob = rng.integers(0, 4, size=(H, W, maxh * maxw)).astype(np.float32)
        ret = dfe.processOutput(geo, T(prob, cuda), True)
        mid = dfe.getMiddleIndex(geo)
        ei, _ = orc.argbest_center(prob, mid, True)
        ey, ex = orc.x2yx(ei, maxh, maxw)
        assert np.array_equal(ret["index"].cpu().numpy(), ei)
        assert np.array_equal(ret["y"].cpu().numpy(), ey) and np.array_equal(ret["x"].cpu().numpy(), ex)
        full = ret["full"].cpu().numpy()
        ho, wo = (40 - H) // 2, (50 - W) // 2
        assert np.array_equal(full[0, ho : ho + H, wo : wo + W], ey.astype(np.float32))
        assert full[:, :ho].sum() == 0 and full[:, :, :wo].sum() == 0
        ty, tx = dfe.x2yx(geo, T(ei, cuda))
        assert np.array_equal(ty.cpu().numpy(), (ei - 1) // maxw + 1) and np.array_equal(tx.cpu().numpy(), (ei - 1) % maxw + 1)


# ------------------------------------------------------------------ golden fixtures
def test_golden_fixtures_on_gpu(dfe, cuda):
    g = np.load(GOLD)
    hK, wK, hW, wW = [int(v) for v in g["kat3_geo"]]
    ctx = dfe.get_ctx(0)
    ctx.set_cost_volume_kernel(1)
    try:
        out = dfe.nn.SSDCostVolume(hW, wW, hK, wK).forward([T(g["kat3_im1"], cuda), T(g["kat3_im2"], cuda)]).cpu().numpy()
    finally:
        ctx.set_cost_volume_kernel(0)
    assert np.array_equal(out, g["kat3_cost"])
    out = dfe.nn.SSDCostVolume(9, 9, 7, 7).forward([T(g["int_f0"], cuda), T(g["int_f1"], cuda)])
    assert ctx.last_kernel() == "ssd_cv_tiled_kernel"
    assert np.array_equal(out.cpu().numpy(), g["int_cost"])
    Ho, Wo = g["int_idx"].shape
    idx = torch.empty((Ho, Wo), dtype=torch.int64, device=cuda)
    ctx.check(dfe.lib().dfe_argbest_center(ctx.handle, out.data_ptr(), Ho * Wo, 81, rp.middle_index(9, 9), 0, idx.data_ptr(), None))
    assert np.array_equal(idx.cpu().numpy(), g["int_idx"])
    ids = torch.arange(1, 161, dtype=torch.int64, device=cuda)
    geo = dict(maxh=8, maxw=8, ratios=[1, 2, 4])
    y, x = dfe.x2yxMulti2(geo, ids)
    assert np.array_equal(y.cpu().numpy(), g["codec_8_8_124_y"]) and np.array_equal(x.cpu().numpy(), g["codec_8_8_124_x"])


@pytest.mark.parametrize(
    "H,W,win,C,thr",
    [
        (96, 128, 33, 3, 0.21),    # row-image kernel, tiles divide the frame
        (90, 100, 33, 1, 0.21),    # row-image kernel on luminance frames
        (75, 80, 33, 3, 0.21),     # last tile row / column shifted inwards; 37 output rows -> 18-row tiles
        (131, 90, 33, 3, 0.11),    # M = 8
        (64, 50, 33, 3, 0.21),     # Wo = 12, Ho = 26
        (90, 110, 17, 3, 0.21),    # 17x17 window: tiled kernel
        (60, 70, 9, 1, 0.21),      # luminance frames
    ],
)
def test_flow_depth_pair_matches_oracle(dfe, cuda, H, W, win, C, thr):
    """dfe_flow_depth_pair_f32 (the bench step): flow / scores centre-pasted into buffers that are NOT pre-zeroed, borders
    zero, depth and confidence from the quirk-preserving cartesian formula -- against the oracle composition."""
    k = 7
    f0, f1, _, (cx, cy) = rp.synth_pair(H, W, C=C, seed=H + win, max_flow=min(win // 2 - 1, 12), noise_sigma=1.0)
    ref = rp.dense_flow_oracle(f0, f1, win, win, k, k, thr=thr)
    ctx = dfe.get_ctx(0)
    t0, t1 = T(f0, cuda), T(f1, cuda)
    flow = torch.full((2, H, W), 7.0, device=cuda)
    scores = torch.full((H, W), -3.0, device=cuda)
    depth = torch.full((H, W), -1.0, device=cuda)
    conf = torch.full((H, W), -1.0, device=cuda)
    ctx.check(dfe.lib().dfe_flow_depth_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, win, win, cx, cy, thr,
                                               flow.data_ptr(), scores.data_ptr(), depth.data_ptr(), conf.data_ptr()))
    name = ctx.last_kernel()
    assert name.startswith("ssd_cv_rowimg_kernel+fused_tail" if win == 33 else "ssd_cv_tiled_kernel+fused_tail"), name
    eflow = ref["flowp"][:2]
    assert np.array_equal(flow.cpu().numpy(), eflow)
    assert np.array_equal(scores.cpu().numpy(), ref["flowp"][3])
    ed, ec = orc.flow_to_depth_cartesian(eflow, cx, cy)
    assert np.allclose(depth.cpu().numpy(), ed, rtol=1e-6, atol=0) and np.array_equal(conf.cpu().numpy(), ec)
    # flow only: depth buffers may be NULL
    flow2 = torch.full((2, H, W), 7.0, device=cuda)
    ctx.check(dfe.lib().dfe_flow_depth_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, win, win, cx, cy, thr,
                                               flow2.data_ptr(), None, None, None))
    assert torch.equal(flow2, flow)


@pytest.mark.parametrize("K,H1,W1,mh,mw", [(3, 9, 11, 5, 4), (30, 20, 26, 7, 9), (1, 6, 70, 17, 17), (8, 33, 5, 1, 1)])
def test_spatial_matching_backward_bit_exact(dfe, cuda, K, H1, W1, mh, mw):
    """N2: nn.SpatialMatching:updateGradInput == oracle (integer-valued data: exact in any order; the kernels also use
    the oracle's (dy, dx) accumulation order, checked on floats)."""
    rng = np.random.default_rng(K + H1)
    for integer in (True, False):
        gen = (lambda s: rng.integers(-5, 6, s).astype(np.float32)) if integer else (lambda s: rng.standard_normal(s).astype(np.float32))
        in1, in2 = gen((K, H1, W1)), gen((K, H1 + mh - 1, W1 + mw - 1))
        go = gen((H1, W1, mh, mw))
        e1, e2 = orc.spatial_matching_backward(in1, in2, go, mh, mw)
        m = dfe.nn.SpatialMatching(mh, mw, False)
        g1, g2 = m.backward([T(in1, cuda), T(in2, cuda)], T(go, cuda))
        assert np.array_equal(g1.cpu().numpy(), e1) and np.array_equal(g2.cpu().numpy(), e2)


def test_radial_matching_backward_bit_exact(dfe, cuda):
    rng = np.random.default_rng(3)
    K, H1, W, hW = 12, 25, 31, 12
    in1, in2 = rng.standard_normal((K, H1, W)).astype(np.float32), rng.standard_normal((K, H1 + hW - 1, W)).astype(np.float32)
    go = rng.standard_normal((H1, W, hW)).astype(np.float32)
    e1, e2 = orc.radial_matching_backward(in1, in2, go, hW)
    g1, g2 = dfe.nn.SpatialRadialMatching(hW).backward([T(in1, cuda), T(in2, cuda)], T(go, cuda))
    assert np.array_equal(g1.cpu().numpy(), e1) and np.array_equal(g2.cpu().numpy(), e2)
    with pytest.raises(ValueError, match="gradOutput"):
        dfe.nn.SpatialRadialMatching(hW).backward([T(in1, cuda), T(in2, cuda)], T(go[:-1], cuda))


# ------------------------------------------------------------------ full size (BASELINE configs[1]): properties
def test_full_vga_cost_volume_properties(dfe, cuda):
    """640x480, C=3, 7x7 patch, 33x33 window: (a) tiled == row-image == reference-order kernel bitwise on integer
    frames (three independent formulations, 1.16 GB each); (b) oracle agreement on a band of rows;
    (c) cost(I,I) is zero exactly at the centre cell; (d) the planted flow is the arg-min."""
    H, W, k, win = 480, 640, 7, 33
    f0, f1, flow, _ = rp.synth_pair(H, W, C=3, seed=0, max_flow=12, noise_sigma=0)
    t0, t1 = T(f0, cuda), T(f1, cuda)
    ctx = dfe.get_ctx(0)
    op = dfe.nn.SSDCostVolume(win, win, k, k)
    ctx.set_cost_volume_kernel(2)
    try:
        tiled = op.forward([t0, t1])
        assert ctx.last_kernel() == "ssd_cv_tiled_kernel"
        ctx.set_cost_volume_kernel(1)
        ref = op.forward([t0, t1])
    finally:
        ctx.set_cost_volume_kernel(0)
    assert tuple(tiled.shape) == (442, 602, 33, 33)
    assert torch.equal(tiled, ref)
    del ref
    rowimg = op.forward([t0, t1])          # auto mode: the row-image kernel, a third formulation of the same sums
    assert ctx.last_kernel() == "ssd_cv_rowimg_kernel"
    assert torch.equal(tiled, rowimg)
    del rowimg
    ctx.set_cost_volume_tile(148)          # the row-image kernel's static 48-row tiles (what the fused build takes at VGA)
    try:
        rowimg = op.forward([t0, t1])
        assert ctx.last_kernel() == "ssd_cv_rowimg_kernel"
    finally:
        ctx.set_cost_volume_tile(0)
    assert torch.equal(tiled, rowimg)
    del rowimg
    rows = (0, 3, 200, 203, 439, 442)
    for r0, r1 in zip(rows[::2], rows[1::2]):
        cpu = np.zeros((442, 602, 33, 33), np.float32)[r0:r1] * 0
        band = orc.ssd_cost_volume(f0[:, r0 : r1 + 38], f1[:, r0 : r1 + 38], k, k, win, win)
        assert np.array_equal(tiled[r0:r1].cpu().numpy(), band)
    mid = rp.middle_index(win, win)
    idx = torch.empty((442, 602), dtype=torch.int64, device=cuda)
    ctx.check(dfe.lib().dfe_argbest_center(ctx.handle, tiled.data_ptr(), idx.numel(), win * win, mid, 0, idx.data_ptr(), None))
    y, x = dfe.x2yx(dict(maxh=win, maxw=win), idx)
    y, x = (y - 17).cpu().numpy(), (x - 17).cpu().numpy()     # 1-based cell -> centred displacement
    inner = (slice(12, 442 - 12), slice(12, 602 - 12))
    py, px = flow[0][19:-19, 19:-19], flow[1][19:-19, 19:-19]
    assert ((y == py) & (x == px))[inner].mean() > 0.95   # flow-field seams and smooth-texture ties aside, the plant is recovered
    # the fused build of the flow pipeline at full size (auto mode) must find exactly these indices and minima
    fidx = torch.empty((442, 602), dtype=torch.int64, device=cuda)
    fbest = torch.empty((442, 602), dtype=torch.float32, device=cuda)
    ffy, ffx = torch.empty_like(fbest), torch.empty_like(fbest)
    ctx.check(dfe.lib().dfe_ssd_flow_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, k, k, win, win, 0.21,
                                        fidx.data_ptr(), fbest.data_ptr(), ffy.data_ptr(), ffx.data_ptr(), None, None))
    assert ctx.last_kernel().startswith("ssd_cv_rowimg_kernel+fused_tail")
    assert torch.equal(fidx, idx)
    assert torch.equal(fbest, tiled.reshape(442 * 602, -1).gather(1, (idx.reshape(-1, 1) - 1)).reshape(442, 602))
    del tiled
    same = op.forward([t1, t1])
    centre = same.reshape(442, 602, -1)[:, :, mid - 1]
    assert float(centre.abs().max()) == 0.0 and float(same.min()) == 0.0


# ------------------------------------------------------------------ real image pair (BASELINE configs[0])
@pytest.mark.parametrize("size,win", [((320, 240), 17), ((640, 480), 33)])
def test_celiu_car_pair_hip_equals_oracle(dfe, cuda, size, win):
    """car1 -> car2 (tests/golden/celiu, the reference's own demo pair): 320x240 with the +-8 window of configs[0] and
    640x480 with the headline +-16 window, through dfe_ssd_flow_f32 (fused build, finalize) and through
    dfe_flow_depth_pair_f32, against the oracle -- uint8-valued frames, so everything is bit-exact."""
    from tests.test_real_image_cpu import load_pair

    f0, f1 = load_pair(size)
    W, H = size
    k = 7
    Ho, Wo = H - k + 1 - win + 1, W - k + 1 - win + 1
    # oracle: cost volume in row chunks (bounded memory), then arg-min with the centre tie-break, decode, extractOutput
    mid = rp.middle_index(win, win)
    eidx = np.empty((Ho, Wo), np.int64)
    ebest = np.empty((Ho, Wo), np.float32)
    esc, eim = np.zeros((Ho, Wo), np.float32), np.zeros((Ho, Wo), np.int64)
    for r0 in range(0, Ho, 64):
        r1 = min(r0 + 64, Ho)
        vol = orc.ssd_cost_volume(f0[:, r0 : r1 + k - 1 + win - 1], f1[:, r0 : r1 + k - 1 + win - 1], k, k, win, win).reshape(r1 - r0, Wo, win * win)
        eidx[r0:r1], ebest[r0:r1] = orc.argbest_center(vol, mid, False)
        orc.extract_output(vol, 0.21, eim[r0:r1], esc[r0:r1])
    ey, ex = orc.x2yx(eidx, win, win)
    ctx = dfe.get_ctx(0)
    t0, t1 = T(f0, cuda), T(f1, cuda)
    idx = torch.empty((Ho, Wo), dtype=torch.int64, device=cuda)
    best, fy, fx, sc = (torch.empty((Ho, Wo), device=cuda) for _ in range(4))
    im = torch.zeros((Ho, Wo), dtype=torch.int64, device=cuda)
    sc.zero_()
    ctx.check(dfe.lib().dfe_ssd_flow_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, k, k, win, win, 0.21, idx.data_ptr(), best.data_ptr(),
                                        fy.data_ptr(), fx.data_ptr(), sc.data_ptr(), im.data_ptr()))
    assert np.array_equal(idx.cpu().numpy(), eidx) and np.array_equal(best.cpu().numpy(), ebest)
    assert np.array_equal(fy.cpu().numpy(), ey.astype(np.float32)) and np.array_equal(fx.cpu().numpy(), ex.astype(np.float32))
    assert np.array_equal(sc.cpu().numpy(), esc) and np.array_equal(im.cpu().numpy(), eim)
    # the pair call: centre-pasted flow, scores, depth
    flow = torch.empty((2, H, W), device=cuda)
    scores, depth, dconf = (torch.empty((H, W), device=cuda) for _ in range(3))
    cx, cy = W / 2 + 17, H / 2 - 9
    ctx.check(dfe.lib().dfe_flow_depth_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, k, win, win, cx, cy, 0.21,
                                               flow.data_ptr(), scores.data_ptr(), depth.data_ptr(), dconf.data_ptr()))
    pt, pl = (H - Ho) // 2, (W - Wo) // 2
    eflow = np.zeros((2, H, W), np.float32)
    eflow[0, pt : pt + Ho, pl : pl + Wo], eflow[1, pt : pt + Ho, pl : pl + Wo] = ey, ex
    assert np.array_equal(flow.cpu().numpy(), eflow)
    ed, ec = orc.flow_to_depth_cartesian(eflow, cx, cy)
    assert np.allclose(depth.cpu().numpy(), ed, rtol=1e-6, atol=0) and np.array_equal(dconf.cpu().numpy(), ec)
    # ... and against the celiu flow the reference ships for this pair (celiu/output/car_flow.jpg decoded by tests/flowcolor.py: direction
    # and relative magnitude; tests/test_real_image_cpu.py has the oracle's twin of this check): the HIP path's confident pixels point the way
    # celiu's flow does, the car blob carries the large leftward displacements
    from tests.test_real_image_cpu import agreement_with_celiu, celiu_field

    U, V, R = celiu_field(half=(size == (320, 240)))
    gflow, gconf = flow.cpu().numpy(), scores.cpu().numpy() > 0
    frac, fx_car, fx_bg, n, ncar = agreement_with_celiu(gflow[0], gflow[1], gconf, U, V, R)
    full = 1 if size == (320, 240) else 2
    assert n > 30000 * full * full and ncar > 3000 * full * full
    assert frac >= 0.70, frac
    assert fx_car < -3.0 * full and -1.0 * full < fx_bg < 1.5 * full, (fx_car, fx_bg)


def test_patch_mode_spatial_matching_equals_oracle(dfe, cuda):
    """tests/test_patches.lua:46-60 on the device: a K x 1 x 1 patch against K x 16 x 16 candidates through
    nn.SpatialMatching(16, 16) -- bit-exact against the oracle (reference summation order), arg-max of -output = the planted class."""
    import math

    rng = np.random.default_rng(3)
    k, maxh = 16, 16
    I0 = rng.random((3, 80, 90), dtype=np.float32)
    c = math.ceil(maxh / 2) - 1
    for fy, fx in ((0, 0), (3, -5), (-7, 8)):
        I1 = np.roll(I0, (fy, fx), axis=(1, 2))
        y, x = 30, 35
        in1 = orc.unfold(I0[:, y : y + k, x : x + k], k, k)
        in2 = orc.unfold(I1[:, y - c : y - c + maxh + k - 1, x - c : x - c + maxh + k - 1], k, k)
        out = dfe.nn.SpatialMatching(maxh, maxh).forward([T(in1, cuda), T(in2, cuda)])
        assert tuple(out.shape) == (1, 1, maxh, maxh)
        assert np.array_equal(out.cpu().numpy(), orc.spatial_matching(in1, in2, maxh, maxh))
        m = int((-out).reshape(-1).argmax().item()) + 1
        assert m == (fy + math.ceil(maxh / 2) - 1) * maxh + fx + math.ceil(maxh / 2)


# ------------------------------------------------------------------ fp16 cost volume (BASELINE configs[4], SURVEY 8(c) numeric contract)
def _half_oracle(cost, scale):
    return (cost * np.float32(scale)).astype(np.float16)


@pytest.mark.parametrize("H,W,C,win,tile", [(96, 130, 3, 33, 0), (131, 90, 3, 33, 118), (75, 47, 1, 33, 0), (70, 90, 3, 31, 0), (60, 64, 3, 9, 0), (50, 60, 2, 5, 0)])
def test_cost_volume_f16_bit_exact_on_integer_frames(dfe, cuda, H, W, C, win, tile):
    """dfe_ssd_cost_volume_f16: half(cost * 2^-8), round to nearest even -- bit-exact against the oracle's fp32 volume converted
    by numpy, for the row-image kernel's fp16 instantiations (33x33 with D constant, 31x31 generic, C = 1 and 3, ragged
    sizes so that partial lines and shifted last tiles occur) and for shapes that take the fp32-bands + convert route."""
    k = 7 if C != 2 else 5
    f0, f1, _, _ = rp.synth_pair(H, W, C=C, seed=H + win, max_flow=4)
    cpu = orc.ssd_cost_volume(f0, f1, k, k, win, win)
    ctx = dfe.get_ctx(0)
    ctx.set_cost_volume_tile(tile)
    try:
        out = torch.full(cpu.shape, -1.0, dtype=torch.float16, device=cuda)
        t0, t1 = T(f0, cuda), T(f1, cuda)
        ctx.check(dfe.lib().dfe_ssd_cost_volume_f16(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, k, win, win, 2.0 ** -8, out.data_ptr()))
        fast = C in (1, 3) and k == 7 and 768 < win * win <= 1096
        assert ctx.last_kernel() == ("ssd_cv_rowimg_kernel_f16" if fast else ctx.last_kernel())
        if fast:
            assert ctx.last_kernel() == "ssd_cv_rowimg_kernel_f16"
    finally:
        ctx.set_cost_volume_tile(0)
    assert np.array_equal(out.cpu().numpy().view(np.uint16), _half_oracle(cpu, 2.0 ** -8).view(np.uint16))


def test_cost_volume_f16_float_frames_within_half_precision(dfe, cuda):
    """Frames in [0, 1] (scale 1): stored values within rel 2^-10 of the fp32 volume, deterministic."""
    f0, f1, _, _ = rp.synth_pair(90, 100, C=3, seed=7, integer=False, max_flow=6)
    cpu = orc.ssd_cost_volume(f0, f1, 7, 7, 33, 33)
    ctx = dfe.get_ctx(0)
    t0, t1 = T(f0, cuda), T(f1, cuda)
    outs = []
    for _ in range(2):
        out = torch.empty(cpu.shape, dtype=torch.float16, device=cuda)
        ctx.check(dfe.lib().dfe_ssd_cost_volume_f16(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, 90, 100, 7, 7, 33, 33, 1.0, out.data_ptr()))
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    g = outs[0].cpu().numpy().astype(np.float32)
    assert (np.abs(g - cpu) <= 2.0 ** -10 * np.abs(cpu) + 2.0 ** -24).all()


@pytest.mark.parametrize("H,W,C", [(96, 130, 3), (480, 640, 3), (75, 47, 1)])
def test_flow_depth_pair_f16_indices_identical_to_f32_path(dfe, cuda, H, W, C):
    """dfe_flow_depth_pair_f16: arg-min before the down-convert -> idx, best, flow, depth identical to the fp32 pipeline's
    (and to the oracle on a small frame); and the volume the step materialises is the fp16 one."""
    f0, f1, _, (cx, cy) = rp.synth_pair(H, W, C=C, seed=H, max_flow=10)
    k, win = 7, 33
    Ho, Wo = H - k - win + 2, W - k - win + 2
    ctx = dfe.get_ctx(0)
    lib = dfe.lib()
    t0, t1 = T(f0, cuda), T(f1, cuda)
    idx32 = torch.empty((Ho, Wo), dtype=torch.int64, device=cuda)
    best32, fy, fx = (torch.empty((Ho, Wo), device=cuda) for _ in range(3))
    ctx.check(lib.dfe_ssd_flow_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, k, win, win, 0.21, idx32.data_ptr(), best32.data_ptr(),
                                   fy.data_ptr(), fx.data_ptr(), None, None))
    flow32 = torch.empty((2, H, W), device=cuda)
    sc, d32, c32 = (torch.empty((H, W), device=cuda) for _ in range(3))
    ctx.check(lib.dfe_flow_depth_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, win, win, cx, cy, 0.21, flow32.data_ptr(), sc.data_ptr(),
                                          d32.data_ptr(), c32.data_ptr()))
    idx16 = torch.empty_like(idx32)
    best16 = torch.empty_like(best32)
    flow16 = torch.empty_like(flow32)
    d16, c16 = torch.empty_like(d32), torch.empty_like(c32)
    ctx.check(lib.dfe_flow_depth_pair_f16(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, win, win, cx, cy, 2.0 ** -8, idx16.data_ptr(),
                                          best16.data_ptr(), flow16.data_ptr(), d16.data_ptr(), c16.data_ptr()))
    assert ctx.last_kernel() == "ssd_cv_rowimg_kernel_f16+fused_tail"
    assert torch.equal(idx16, idx32) and torch.equal(best16, best32)
    assert torch.equal(flow16, flow32) and torch.equal(d16, d32) and torch.equal(c16, c32)
    if H < 200:
        ref = rp.dense_flow_oracle(f0, f1, win, win, k, k)
        assert np.array_equal(idx16.cpu().numpy(), ref["idx"]) and np.array_equal(best16.cpu().numpy(), ref["best"])
    with pytest.raises(dfe.DfeError):   # no fused fp16 kernel for a 9x9 window: the caller takes the fp32 path
        ctx.check(lib.dfe_flow_depth_pair_f16(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, 9, 9, cx, cy, 2.0 ** -8, None, None, flow16.data_ptr(), None, None))


def test_flow_depth_pair_f16_multiband_equals_oracle(dfe, cuda):
    """dfe_flow_depth_pair_f16 with the scratch arena limited so that the fp16 volume is built in several row bands (the path the
    4K workload takes: two bands under the default 16-GiB limit) -- idx / best against the oracle, flow / depth against the
    one-band fp32 pipeline, and the band split itself: balanced, so no short last band is left for which the fp16 pipeline
    would have no kernel."""
    H, W, C, k, win = 131, 140, 3, 7, 33
    f0, f1, _, (cx, cy) = rp.synth_pair(H, W, C=C, seed=21, max_flow=10)
    Ho, Wo = H - k - win + 2, W - k - win + 2
    ref = rp.dense_flow_oracle(f0, f1, win, win, k, k)
    ctx = dfe.get_ctx(0)
    lib = dfe.lib()
    t0, t1 = T(f0, cuda), T(f1, cuda)
    flow32 = torch.empty((2, H, W), device=cuda)
    sc, d32, c32 = (torch.empty((H, W), device=cuda) for _ in range(3))
    ctx.check(lib.dfe_flow_depth_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, win, win, cx, cy, 0.21, flow32.data_ptr(), sc.data_ptr(),
                                          d32.data_ptr(), c32.data_ptr()))
    vol16 = Ho * Wo * win * win * 2
    for limit in (vol16 // 3, vol16 // 7, vol16 // 12):      # 4, 8 and 14 bands (the last: bands of 6-7 rows, one minimal tile each)
        ctx.check(lib.dfe_set_scratch_limit(ctx.handle, max(limit, 1 << 20)))
        try:
            idx = torch.full((Ho, Wo), -1, dtype=torch.int64, device=cuda)
            best = torch.full((Ho, Wo), -1.0, device=cuda)
            flow = torch.full((2, H, W), -7.0, device=cuda)
            dd, cc = torch.full((H, W), -7.0, device=cuda), torch.full((H, W), -7.0, device=cuda)
            ctx.check(lib.dfe_profile_enable(ctx.handle, 1))
            ctx.check(lib.dfe_flow_depth_pair_f16(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, win, win, cx, cy, 2.0 ** -8, idx.data_ptr(),
                                                  best.data_ptr(), flow.data_ptr(), dd.data_ptr(), cc.data_ptr()))
            ms, n = C_double(), C_int()
            ctx.check(lib.dfe_profile_read(ctx.handle, byref(ms), byref(n)))
            assert n.value >= 3, "expected a multi-band build, got %d launch(es)" % n.value
            assert ctx.last_kernel() == "ssd_cv_rowimg_kernel_f16+fused_tail"
        finally:
            ctx.check(lib.dfe_profile_enable(ctx.handle, 0))
            ctx.check(lib.dfe_set_scratch_limit(ctx.handle, 16 << 30))
        assert np.array_equal(idx.cpu().numpy(), ref["idx"]) and np.array_equal(best.cpu().numpy(), ref["best"])
        assert torch.equal(flow, flow32) and torch.equal(dd, d32) and torch.equal(cc, c32)


def test_4k_f16_two_bands_equals_f32_path(dfe, cuda):
    """BASELINE configs[4] at its size, single scale: 3840x2160 / 33x33 through dfe_flow_depth_pair_f16 (17.8 GB of fp16 volume:
    two bands under the default limit) == dfe_flow_depth_pair_f32 (35.5 GB of fp32 volume: three bands) on flow, depth and
    confidence; idx / best == dfe_ssd_flow_f32's; the planted flow is recovered."""
    H, W, C, k, win = 2160, 3840, 3, 7, 33
    f0, f1, pflow, (cx, cy) = rp.synth_pair(H, W, C=C, seed=1, max_flow=12, noise_sigma=0)
    Ho, Wo = H - k - win + 2, W - k - win + 2
    ctx = dfe.get_ctx(0)
    lib = dfe.lib()
    t0, t1 = T(f0, cuda), T(f1, cuda)
    idx16 = torch.empty((Ho, Wo), dtype=torch.int64, device=cuda)
    best16 = torch.empty((Ho, Wo), device=cuda)
    flow16 = torch.empty((2, H, W), device=cuda)
    d16, c16 = torch.empty((H, W), device=cuda), torch.empty((H, W), device=cuda)
    ctx.check(lib.dfe_profile_enable(ctx.handle, 1))
    try:
        ctx.check(lib.dfe_flow_depth_pair_f16(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, win, win, cx, cy, 2.0 ** -8, idx16.data_ptr(),
                                              best16.data_ptr(), flow16.data_ptr(), d16.data_ptr(), c16.data_ptr()))
        ms, n = C_double(), C_int()
        ctx.check(lib.dfe_profile_read(ctx.handle, byref(ms), byref(n)))
    finally:
        ctx.check(lib.dfe_profile_enable(ctx.handle, 0))
    assert n.value == 2 and ctx.last_kernel() == "ssd_cv_rowimg_kernel_f16+fused_tail"
    flow32 = torch.empty((2, H, W), device=cuda)
    sc, d32, c32 = (torch.empty((H, W), device=cuda) for _ in range(3))
    ctx.check(lib.dfe_flow_depth_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, win, win, cx, cy, 0.21, flow32.data_ptr(), sc.data_ptr(),
                                          d32.data_ptr(), c32.data_ptr()))
    assert torch.equal(flow16, flow32) and torch.equal(d16, d32) and torch.equal(c16, c32)
    idx32 = torch.empty_like(idx16)
    best32 = torch.empty_like(best16)
    ctx.check(lib.dfe_ssd_flow_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), C, H, W, k, k, win, win, 0.21, idx32.data_ptr(), best32.data_ptr(),
                                   None, None, None, None))
    assert torch.equal(idx16, idx32) and torch.equal(best16, best32)
    fy, fx = flow16[0].cpu().numpy(), flow16[1].cpu().numpy()
    inner = (slice(40, H - 40), slice(40, W - 40))
    assert ((fy == pflow[0]) & (fx == pflow[1]))[inner].mean() > 0.95


@pytest.mark.parametrize("H,W", [(720, 1280), (1080, 1920)])
def test_benched_single_scale_sizes_properties(dfe, cuda, H, W):
    """The 720p / 1080p workloads of bench.py (33x33, C=3), where the persistent sweep's cut, the aligned-front schedule and the
    band logic take other branches than at VGA.  Size-independent properties: (a) the auto (row-image, swept) build == the
    tiled kernel == the oracle on row bands at the top, across the middle and at the bottom, bit for bit (integer frames);
    (b) the fused pipeline's idx / best == dfe_argbest_center over the materialised volume; (c) the planted flow is the
    arg-min; (d) dfe_flow_depth_pair_f32 (the bench step) agrees with (b) and zeroes its border."""
    k, win = 7, 33
    f0, f1, pflow, (cx, cy) = rp.synth_pair(H, W, C=3, seed=3, max_flow=12, noise_sigma=0)
    Ho, Wo = H - k - win + 2, W - k - win + 2
    t0, t1 = T(f0, cuda), T(f1, cuda)
    ctx = dfe.get_ctx(0)
    lib = dfe.lib()
    op = dfe.nn.SSDCostVolume(win, win, k, k)
    vol = op.forward([t0, t1])
    assert ctx.last_kernel() == "ssd_cv_rowimg_kernel" and tuple(vol.shape) == (Ho, Wo, win, win)
    nb = 20
    for r0 in (0, Ho // 3 - 7, Ho // 2 - 10, Ho - nb):
        rows = slice(r0, r0 + nb + k - 1 + win - 1)
        b0, b1 = np.ascontiguousarray(f0[:, rows]), np.ascontiguousarray(f1[:, rows])
        ctx.set_cost_volume_kernel(2)
        try:
            band = op.forward([T(b0, cuda), T(b1, cuda)])
            assert ctx.last_kernel() == "ssd_cv_tiled_kernel"
        finally:
            ctx.set_cost_volume_kernel(0)
        assert torch.equal(band, vol[r0 : r0 + nb]), "rows %d.." % r0
        cpu = orc.ssd_cost_volume(b0[:, : 3 + k - 1 + win - 1], b1[:, : 3 + k - 1 + win - 1], k, k, win, win)
        assert np.array_equal(vol[r0 : r0 + 3].cpu().numpy(), cpu)
    mid = rp.middle_index(win, win)
    idx = torch.empty((Ho, Wo), dtype=torch.int64, device=cuda)
    best = torch.empty((Ho, Wo), device=cuda)
    ctx.check(lib.dfe_argbest_center(ctx.handle, vol.data_ptr(), idx.numel(), win * win, mid, 0, idx.data_ptr(), best.data_ptr()))
    del vol
    fidx, fbest = torch.empty_like(idx), torch.empty_like(best)
    ffy, ffx = torch.empty_like(best), torch.empty_like(best)
    ctx.check(lib.dfe_ssd_flow_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, k, k, win, win, 0.21, fidx.data_ptr(), fbest.data_ptr(),
                                   ffy.data_ptr(), ffx.data_ptr(), None, None))
    assert ctx.last_kernel().startswith("ssd_cv_rowimg_kernel+fused_tail")
    assert torch.equal(fidx, idx) and torch.equal(fbest, best)
    y, x = dfe.x2yx(dict(maxh=win, maxw=win), idx)
    assert torch.equal((y - 17).float(), ffy) and torch.equal((x - 17).float(), ffx)
    inner = (slice(12, Ho - 12), slice(12, Wo - 12))
    py, px = pflow[0][19:-19, 19:-19], pflow[1][19:-19, 19:-19]
    assert ((ffy.cpu().numpy() == py) & (ffx.cpu().numpy() == px))[inner].mean() > 0.95
    flow = torch.full((2, H, W), -9.0, device=cuda)
    sc, dd, cc = (torch.full((H, W), -9.0, device=cuda) for _ in range(3))
    ctx.check(lib.dfe_flow_depth_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, k, win, win, cx, cy, 0.21, flow.data_ptr(), sc.data_ptr(),
                                          dd.data_ptr(), cc.data_ptr()))
    assert torch.equal(flow[0, 19:-19, 19:-19], ffy) and torch.equal(flow[1, 19:-19, 19:-19], ffx)
    assert float(flow[:, :19].abs().max()) == 0 and float(flow[:, :, -19:].abs().max()) == 0
    ed, ec = orc.flow_to_depth_cartesian(flow.cpu().numpy(), cx, cy)
    assert np.allclose(dd.cpu().numpy(), ed, rtol=1e-6, atol=0) and np.array_equal(cc.cpu().numpy(), ec)


@pytest.mark.parametrize("name,H,W,bands", [("single", 200, 256, 3), ("pyramid", 288, 512, 3), ("single", 2160, 3840, 8), ("pyramid", 2160, 3840, 8)])
def test_row_band_split_equals_whole_frame(dfe, cuda, name, H, W, bands):
    """BASELINE configs[4] over N GPUs = ONE pair split into row bands (bench.py's *-bands workloads; the N-rank protocol runs on gloo in
    tests/test_dist_cpu.py).  Here the band arithmetic on the real pipelines, on one GPU: the bands a rank would compute -- owned rows
    + halo, its own scratch, the epipole shifted by the band's first row -- stitched together equal the whole frame computed in one
    call, bit for bit: dfe_flow_depth_pair_f16 (flow + depth) and the 5-level fp16 pyramid, at a small size and at 3840x2160 / 8 bands."""
    import bench

    ratios = (1, 2, 4, 8, 16)
    Cc, k = 3, 7
    win = 33 if name == "single" else 8
    f0, f1, _, (cx, cy) = rp.synth_pair(H, W, C=Cc, seed=7, max_flow=10)
    frames = torch.from_numpy(np.stack([f0, f1]).astype(np.uint8)).to(cuda)
    ctx = dfe.get_ctx(0)
    compute = bench.make_band_compute(name, dfe, ctx, cuda, Cc, W, k, win, ratios, cx, cy)
    whole = compute(frames, 0, H, 0)
    plan = bench.band_plan(H, bands, compute.align, compute.halo)
    assert len(plan) == bands
    full = bench.run_banded_step(frames, plan, 1, 0, None, compute, cuda, torch.cuda.synchronize)
    for a, b in zip(full, whole):
        assert a.shape == b.shape and torch.equal(a, b)
    if name == "single":      # and the band pipeline is the product pipeline: same flow as the fp32 one-call on the float frames
        flow = torch.empty((2, H, W), device=cuda)
        sc, dd, cc = (torch.empty((H, W), device=cuda) for _ in range(3))
        t0, t1 = T(f0, cuda), T(f1, cuda)
        ctx.check(dfe.lib().dfe_flow_depth_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), Cc, H, W, k, win, win, cx, cy, 0.21, flow.data_ptr(), sc.data_ptr(),
                                                    dd.data_ptr(), cc.data_ptr()))
        assert torch.equal(full[0], flow.to(torch.int16)) and torch.equal(full[1], dd)
    # a halo one row short is NOT enough (the test would not notice a halo that is merely generous)
    short = bench.band_plan(H, bands, compute.align, compute.halo - compute.align)
    bad = bench.run_banded_step(frames, short, 1, 0, None, compute, cuda, torch.cuda.synchronize)
    assert not all(torch.equal(a, b) for a, b in zip(bad, whole))


def test_stage_timers_with_the_reference_names(dfe, cuda):
    """dfe_stage_timers_*: load / filter / match / extract (depth_estimation_opticalflow.lua:144-148) around the launches of the
    one-call pipelines: every stage a pipeline has shows up with a positive time, nothing is counted twice (the stages sum to
    less than the wall time of the calls), read resets."""
    import time
    from ctypes import c_double, c_int

    ctx = dfe.get_ctx(0)
    lib = dfe.lib()
    H, W = 240, 320
    f0, f1, _, (cx, cy) = rp.synth_pair(H, W, C=3, seed=1, max_flow=8)
    t0, t1 = T(f0, cuda), T(f1, cuda)
    flow = torch.empty((2, H, W), device=cuda)
    sc, dd, cc = (torch.empty((H, W), device=cuda) for _ in range(3))

    def read():
        ms, n = (c_double * 4)(), (c_int * 4)()
        ctx.check(lib.dfe_stage_timers_read(ctx.handle, ms, n))
        return list(ms), list(n)

    ctx.check(lib.dfe_stage_timers_enable(ctx.handle, 1))
    try:
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            ctx.check(lib.dfe_flow_depth_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, 7, 33, 33, cx, cy, 0.21, flow.data_ptr(), sc.data_ptr(),
                                                  dd.data_ptr(), cc.data_ptr()))
        ms, n = read()
        wall = (time.perf_counter() - t) * 1e3
        assert n == [0, 0, 3, 3] and ms[2] > 0 and ms[3] > 0 and ms[0] == 0 and ms[1] == 0
        assert ms[2] > ms[3] and sum(ms) <= wall                      # the build dominates; no double counting
        assert read() == ([0.0] * 4, [0] * 4)                         # read resets
        # the learned multiscale matcher has a filter stage, its arg-max is fused into the cascade (match)
        geo = dict(maxh=8, maxw=8, ratios=[1, 2, 4], multiscale=True, layers=[(3, 5, 5, 4), (4, 5, 5, 10)], share_filters=True, hImg=H, wImg=W,
                   output_extraction_method="max")
        model = dfe.getModelMultiscale(geo, True, False, device=cuda, generator=torch.Generator().manual_seed(1))
        model.forwardFlow([t0 / 255, t1 / 255], False, one_call=True)
        ms, n = read()
        assert n[1] == 1 + 2 and n[2] == 1 + 1 and n[3] == 0 and ms[1] > 0 and ms[2] > 0      # prep + one batched launch per layer; the batched matcher + the cascade
        # a staged upload counts as load
        host = np.zeros(1 << 16, np.float32)
        dev = torch.empty(1 << 16, device=cuda)
        ctx.check(lib.dfe_memcpy_h2d(ctx.handle, dev.data_ptr(), host.ctypes.data, host.nbytes))
        ms, n = read()
        assert n == [1, 0, 0, 0] and ms[0] > 0
    finally:
        ctx.check(lib.dfe_stage_timers_enable(ctx.handle, 0))
    ctx.check(lib.dfe_flow_depth_pair_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, 7, 33, 33, cx, cy, 0.21, flow.data_ptr(), sc.data_ptr(),
                                          dd.data_ptr(), cc.data_ptr()))
    assert read()[1] == [0] * 4                                       # off: nothing recorded


# ------------------------------------------------------------------ one device per ctx
def test_entry_points_run_on_their_ctx_device_and_leave_the_callers_alone(dfe, cuda):
    """Every entry point switches to its ctx's device for the call and restores the caller's current device (DFE_ENTER); two
    ctxs in one host thread give the same result.  With a second GPU present, a ctx on device 1 is driven while device 0 is
    current (and the other way round)."""
    import ctypes as C

    lib = dfe.lib()
    hip = C.CDLL("libamdhip64.so")
    cur = C.c_int(-1)

    def current():
        assert hip.hipGetDevice(C.byref(cur)) == 0
        return cur.value

    ndev = torch.cuda.device_count()
    f0, f1, _, _ = rp.synth_pair(60, 64, C=3, seed=3)
    ref = None
    for dev in range(min(ndev, 2)):
        before = current()
        h = C.c_void_p()
        assert lib.dfe_ctx_create(dev, None, 1, C.byref(h)) == 0
        assert current() == before, "dfe_ctx_create changed the caller's device"
        d = torch.device("cuda", dev)
        t0, t1 = T(f0, d), T(f1, d)
        out = torch.empty((60 - 7 - 8 + 2, 64 - 7 - 8 + 2, 8, 8), device=d)
        other = (dev + 1) % ndev
        assert hip.hipSetDevice(other) == 0            # the caller's device is NOT the ctx's (when there are two)
        rc = lib.dfe_ssd_cost_volume_f32(h, t0.data_ptr(), t1.data_ptr(), 3, 60, 64, 7, 7, 8, 8, out.data_ptr())
        assert rc == 0, lib.dfe_last_error(h)
        assert current() == other, "an entry point left the device switched"
        torch.cuda.synchronize(d)
        got = out.cpu().numpy()
        if ref is None:
            ref = got
            np.testing.assert_array_equal(got, orc.ssd_cost_volume(f0, f1, 7, 7, 8, 8))
        else:
            np.testing.assert_array_equal(got, ref)
        lib.dfe_ctx_destroy(h)
        assert hip.hipSetDevice(before) == 0


# ------------------------------------------------------------------ version2/: the single-scale learned model
@pytest.mark.parametrize("H,W,layers,win", [
    (60, 300, [(3, 17, 17, 8)], 17),                 # the script's geometry with fewer planes: 17 x 17 kernel, 17 x 17 window (flat-tile matcher, extra row)
    (50, 330, [(3, 9, 9, 5)], 17),                  # ragged: W1 = 306 (not a multiple of 4), H1 = 26
    (48, 64, [(3, 5, 5, 4), (4, 3, 3, 6)], 9),      # two layers, small window (chunk matcher)
])
def test_version2_one_call_equals_staged_and_oracle(dfe, cuda, H, W, layers, win):
    """version2/test.lua:40-51: getNetwork(datap), flat parameters loaded, forward on a frame pair, the dense decode.
    dfe_version2_flow_pair_f32 == the module-by-module host path bit for bit (volume, index, flows); both against the oracle
    composition (tests/refpath.version2_flow_oracle): the volume to 1e-4 relative (the normalisation divides by a device sqrt), the
    arg-min wherever the oracle's two smallest costs are not within that band."""
    v2 = dfe.version2
    datap = v2.defaultDatap(wImg=W, hImg=H, normalization_k=9, layers=layers, wWin=win, hWin=win)
    g = torch.Generator().manual_seed(H + W)
    net = v2.getNetwork(datap, device=cuda, generator=g)
    flat = (torch.rand(net.flatParameters().numel(), generator=g) - 0.5) * 0.2
    net.loadParameters(flat)                                               # parameters:copy(torch.load(...))
    assert torch.equal(net.flatParameters().cpu(), flat)
    f1c = net.modules[0].modules[0].modules[2:]
    f2c = net.modules[0].modules[1].modules[1:]
    assert all(a.weight.data_ptr() == b.weight.data_ptr() for a, b in zip(f1c, f2c)), "the second branch shares the first one's weights"
    rng = np.random.default_rng(W)
    prev = rng.random((3, H, W), dtype=np.float32)
    cur = np.roll(prev, (2, -3), axis=(1, 2)) + rng.normal(0, 0.01, (3, H, W)).astype(np.float32)
    tp, tc = T(prev, cuda), T(cur, cuda)
    staged = v2.flowPair(net, datap, tp, tc, one_call=False, want_volume=True)
    one = v2.flowPair(net, datap, tp, tc, one_call=True, want_volume=True)
    for k in ("volume", "index", "xflow", "yflow"):
        assert torch.equal(staged[k], one[k]), k
    lean = v2.flowPair(net, datap, tp, tc, one_call=True, want_volume=False)   # nobody reads the volume: matcher + first-min decode in one kernel
    if win in (16, 17) and W - (win - 1) - (datap["wKernel"] - 1) >= 253:
        assert dfe.get_ctx(0).last_kernel() == "feat_matching_flat_kernel+argmin"
    for k in ("index", "xflow", "yflow"):
        assert torch.equal(lean[k], one[k]), k
    xf, yf = v2.decodeFlow(staged["volume"], datap)                        # test.lua:45-51 on the module output
    assert torch.equal(xf.to(torch.float32), one["xflow"]) and torch.equal(yf.to(torch.float32), one["yflow"])
    ws = [m.weight.cpu().numpy() for m in f1c]
    bs = [m.bias.cpu().numpy() for m in f1c]
    ref = rp.version2_flow_oracle(prev, cur, datap, ws, bs)
    vol = one["volume"].cpu().numpy()
    assert vol.shape == ref["volume"].shape
    assert np.allclose(vol, ref["volume"], rtol=1e-4, atol=1e-5 * float(np.abs(ref["volume"]).max()))
    srt = np.sort(ref["volume"].reshape(vol.shape[0], vol.shape[1], -1), axis=2)
    clear = (srt[..., 1] - srt[..., 0]) > 2e-4 * srt[..., 1] + 1e-5 * float(np.abs(ref["volume"]).max())
    assert clear.mean() > 0.9
    assert np.array_equal(one["index"].cpu().numpy()[clear], ref["index"][clear])
    assert np.array_equal(one["xflow"].cpu().numpy()[clear], ref["xflow"][clear]) and np.array_equal(one["yflow"].cpu().numpy()[clear], ref["yflow"][clear])
    # the planted shift is what most pixels find: cur = prev rolled by (+2, -3) -> the window cell of prev's pixel in cur
    inner = one["yflow"].cpu().numpy()[8:-8, 8:-8], one["xflow"].cpu().numpy()[8:-8, 8:-8]
    assert np.mean((inner[0] == 2) & (inner[1] == -3)) > 0.6


def test_version2_trainer_network_patch_mode(dfe, cuda):
    """getTrainerNetwork (version2/network.lua:41-47) in patch mode: a hKernel x wKernel patch against its (hKernel + hWin - 1)-sized
    neighbourhood -> log-probabilities over the hWin * wWin displacements; exp sums to one, the arg-max is the planted displacement."""
    v2 = dfe.version2
    datap = v2.defaultDatap(normalization_k=5, layers=[(3, 5, 5, 6)], wWin=9, hWin=9)
    g = torch.Generator().manual_seed(5)
    net = v2.getTrainerNetwork(datap, device=cuda, generator=g)
    # patch mode bypasses the crop (the trainer feeds already-cropped patches): replace it by the identity, as the patches are pre-cut
    net.modules[0].modules[0].modules[1] = dfe.radial.SpatialPadding(0, 0, 0, 0)
    rng = np.random.default_rng(9)
    big = rng.random((3, 5 + 8, 5 + 8), dtype=np.float32)
    dy, dx = 6, 2
    patch = big[:, dy : dy + 5, dx : dx + 5].copy()
    out = net.forward([T(patch, cuda), T(big, cuda)])
    assert tuple(out.shape) == (81,)
    assert abs(float(out.exp().sum()) - 1.0) < 1e-4
    # (the normalisation sees different neighbourhoods in the two branches, so the match is approximate: the planted cell is among the best)
    assert int(out.argmax()) in {dy * 9 + dx + o for o in (-10, -9, -8, -1, 0, 1, 8, 9, 10)}


# ------------------------------------------------------------------ prepareInput / rgb2y / uint8 ingest
def test_rgb2y_and_prepare_input(dfe, cuda):
    """opticalflow_model.lua:131-151: luminance switch for one-plane models (dfe_rgb2y_f32 == oracle bit for bit), the single-scale
    narrow of patch 1 by the search window (rows / columns ceil(max/2) .., H - maxh + 1 of them), the multiscale pass-through, the
    prefilter plane check, and the line as literally written (both patches = the luminance of patch 1)."""
    rng = np.random.default_rng(4)
    a = rng.random((3, 40, 52), dtype=np.float32)
    b = rng.random((3, 40, 52), dtype=np.float32)
    ta, tb = T(a, cuda), T(b, cuda)
    assert np.array_equal(dfe.rgb2y(ta).cpu().numpy(), orc.rgb2y(a))
    geo = dict(layers=[[1, 5, 5, 4]], maxh=16, maxw=17, multiscale=False)
    p1, p2 = dfe.prepareInput(geo, ta, tb)
    assert tuple(p2.shape) == (1, 40, 52) and tuple(p1.shape) == (1, 40 - 16 + 1, 52 - 17 + 1)
    assert np.array_equal(p2.cpu().numpy(), orc.rgb2y(b))
    assert np.array_equal(p1.cpu().numpy(), orc.rgb2y(a)[:, 7 : 7 + 25, 8 : 8 + 36])          # narrow(2, ceil(16/2) = 8, 25) 1-based -> rows 7..31
    l1, l2 = dfe.prepareInput(geo, ta, tb, literal_rgb2y=True)
    assert np.array_equal(l2.cpu().numpy(), orc.rgb2y(a))                                      # as written: patch2 is overwritten by patch1's luminance
    m1, m2 = dfe.prepareInput(dict(layers=[[3, 5, 5, 4]], maxh=8, maxw=8, multiscale=True), ta, tb)
    assert m1 is ta and m2 is tb
    with pytest.raises(AssertionError):
        dfe.prepareInput(dict(layers=[[3, 5, 5, 4]], maxh=8, maxw=8, multiscale=True, prefilter=True), ta, tb)   # 3 planes, the stack ends in 4
    with pytest.raises(AssertionError):
        dfe.prepareInput(geo, ta, tb[:, :-1])


def test_uint8_frames_equal_the_fp32_entries(dfe, cuda):
    """dfe_flow_depth_pair_u8 / dfe_multiscale_flow_pair_u8 on uint8 frames == the fp32 entries on float(frame) * scale, bit for bit
    (odd sizes: the conversion's 4-pixel and byte paths), and dfe_u8_to_f32 itself."""
    from depth_estimation_amd._lib import ratios_array

    ctx = dfe.get_ctx(0)
    lib = dfe.lib()
    rng = np.random.default_rng(12)
    H, W, k, win = 75, 101, 7, 33
    u0 = rng.integers(0, 256, (3, H, W), dtype=np.uint8)
    u1 = np.roll(u0, (1, -2), axis=(1, 2))
    t0, t1 = torch.from_numpy(u0).to(cuda), torch.from_numpy(u1).to(cuda)
    for scale in (1.0, 1.0 / 255.0):
        f0 = torch.empty((3, H, W), device=cuda)
        ctx.check(lib.dfe_u8_to_f32(ctx.handle, t0.data_ptr(), t0.numel(), scale, f0.data_ptr()))
        assert np.array_equal(f0.cpu().numpy(), u0.astype(np.float32) * np.float32(scale))
        odd = torch.empty((t0.numel() - 1,), device=cuda)
        ctx.check(lib.dfe_u8_to_f32(ctx.handle, t0.data_ptr() + 1, t0.numel() - 1, scale, odd.data_ptr()))   # unaligned source: the byte path
        assert np.array_equal(odd.cpu().numpy(), u0.reshape(-1)[1:].astype(np.float32) * np.float32(scale))
    f0, f1 = t0.to(torch.float32), t1.to(torch.float32)
    outs = []
    for which in ("f32", "u8"):
        flow = torch.full((2, H, W), -9.0, device=cuda)
        scores, depth, conf = (torch.full((H, W), -9.0, device=cuda) for _ in range(3))
        if which == "f32":
            ctx.check(lib.dfe_flow_depth_pair_f32(ctx.handle, f0.data_ptr(), f1.data_ptr(), 3, H, W, k, win, win, 50.0, 37.0, 0.21, flow.data_ptr(), scores.data_ptr(),
                                                  depth.data_ptr(), conf.data_ptr()))
        else:
            ctx.check(lib.dfe_flow_depth_pair_u8(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, k, win, win, 50.0, 37.0, 0.21, 1.0, flow.data_ptr(), scores.data_ptr(),
                                                 depth.data_ptr(), conf.data_ptr()))
        outs.append((flow, scores, depth, conf))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    Hm, Wm = 64, 96
    m0, m1 = t0[:, :Hm, :Wm].contiguous(), t1[:, :Hm, :Wm].contiguous()
    rr, n = ratios_array([1, 2, 4])
    res = []
    for which, fs in (("f32", 0.0), ("u8", 0.0), ("f16", 2.0 ** -8), ("u8", 2.0 ** -8)):
        flow = torch.empty((2, Hm, Wm), device=cuda)
        idx = torch.empty((Hm, Wm), dtype=torch.int64, device=cuda)
        a0, a1 = m0.to(torch.float32), m1.to(torch.float32)
        if which == "f32":
            ctx.check(lib.dfe_multiscale_flow_pair_f32(ctx.handle, a0.data_ptr(), a1.data_ptr(), 3, Hm, Wm, 7, 8, 8, rr, n, flow.data_ptr(), idx.data_ptr()))
        elif which == "f16":
            ctx.check(lib.dfe_multiscale_flow_pair_f16(ctx.handle, a0.data_ptr(), a1.data_ptr(), 3, Hm, Wm, 7, 8, 8, rr, n, fs, flow.data_ptr(), idx.data_ptr()))
        else:
            ctx.check(lib.dfe_multiscale_flow_pair_u8(ctx.handle, m0.data_ptr(), m1.data_ptr(), 3, Hm, Wm, 7, 8, 8, rr, n, 1.0, fs, flow.data_ptr(), idx.data_ptr()))
        res.append((flow, idx))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[2][0], res[3][0]) and torch.equal(res[2][1], res[3][1])


def test_uint8_pyramid_large_frame_equals_fp32_entry(dfe, cuda):
    """dfe_multiscale_flow_pair_u8 where every scale is made from ONE read of the frames (prep_tiles_kernel, frames of 1.5 MP and more): the
    bytes go into the preparation kernel as they are -- float(byte) * scale at the load -- and the result is that of the fp32 entry on the
    converted frames, bit for bit, with a scale that is not a power of two."""
    from depth_estimation_amd._lib import ratios_array

    ctx = dfe.get_ctx(0)
    lib = dfe.lib()
    rng = np.random.default_rng(21)
    H, W = 1024, 1536
    u0 = rng.integers(0, 256, (3, H, W), dtype=np.uint8)
    u1 = np.roll(u0, (2, -3), axis=(1, 2))
    t0, t1 = torch.from_numpy(u0).to(cuda), torch.from_numpy(u1).to(cuda)
    scale = 1.0 / 255.0
    f0, f1 = torch.empty((3, H, W), device=cuda), torch.empty((3, H, W), device=cuda)
    ctx.check(lib.dfe_u8_to_f32(ctx.handle, t0.data_ptr(), t0.numel(), scale, f0.data_ptr()))
    ctx.check(lib.dfe_u8_to_f32(ctx.handle, t1.data_ptr(), t1.numel(), scale, f1.data_ptr()))
    rr, n = ratios_array([1, 2, 4, 8])
    res = []
    for which in ("f32", "u8"):
        flow = torch.empty((2, H, W), device=cuda)
        idx = torch.empty((H, W), dtype=torch.int64, device=cuda)
        if which == "f32":
            ctx.check(lib.dfe_multiscale_flow_pair_f32(ctx.handle, f0.data_ptr(), f1.data_ptr(), 3, H, W, 7, 8, 8, rr, n, flow.data_ptr(), idx.data_ptr()))
        else:
            ctx.check(lib.dfe_multiscale_flow_pair_u8(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, 7, 8, 8, rr, n, scale, 0.0, flow.data_ptr(), idx.data_ptr()))
        res.append((flow, idx))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    # the planted shift is found in the interior
    fl = res[1][0].cpu().numpy()
    assert (fl[0, 100:-100, 100:-100] == 2).mean() > 0.9 and (fl[1, 100:-100, 100:-100] == -3).mean() > 0.9


def test_time_matching_lua_exact_shape(dfe, cuda):
    """tests/time_matching.lua:5-47 at its own shape: getFilter({3,5,5,4},{4,5,5,4},{4,5,5,10}) on randn 3 x 180 x 320 frames, the narrow of
    prepareInput, nn.SpatialMatching(16, 16) on the 10-plane 168 x 308 features (output 153 x 293 x 16 x 16) and the script's min over
    the leading dimension of its Reshape(256, 293, 153) view -- features and volume bit-exact against the oracle (the exact convolution and
    the k-ordered matcher), the minimum against numpy."""
    g = torch.Generator().manual_seed(1)
    geometry = dict(maxh=16, maxw=16, layers=[[3, 5, 5, 4], [4, 5, 5, 4], [4, 5, 5, 10]], multiscale=False, prefilter=True)
    filt = dfe.getFilter(geometry, device=cuda, generator=g)
    im1, im2 = torch.randn((3, 180, 320), generator=g), torch.randn((3, 180, 320), generator=g)
    f1 = filt.forward(im1.to(cuda)).clone()
    f2 = filt.forward(im2.to(cuda))
    assert tuple(f1.shape) == (10, 168, 308)
    layers = []
    for m in filt.modules:
        if isinstance(m, dfe.network.Tanh):
            layers[-1]["tanh"] = True
        else:
            layers.append({"weight": m.weight.cpu().numpy(), "bias": m.bias.cpu().numpy(), "tanh": False})
    r1, r2 = rp.filter_stack_oracle(im1.numpy(), layers), rp.filter_stack_oracle(im2.numpy(), layers)
    assert np.allclose(f1.cpu().numpy(), r1, rtol=0, atol=2e-6) and np.allclose(f2.cpu().numpy(), r2, rtol=0, atol=2e-6)   # (device tanhf vs glibc)
    p1, p2 = dfe.prepareInput(geometry, f1, f2)
    assert tuple(p1.shape) == (10, 153, 293)
    out = dfe.nn.SpatialMatching(16, 16, False).forward([p1, p2])
    assert dfe.get_ctx(0).last_kernel() == "feat_matching_flat_kernel"
    assert np.array_equal(out.cpu().numpy(), orc.spatial_matching(p1.cpu().numpy(), p2.cpu().numpy(), 16, 16))
    ctx = dfe.get_ctx(0)
    M = 293 * 153
    mn = torch.empty((M,), device=cuda)
    mi = torch.empty((M,), dtype=torch.int64, device=cuda)
    ctx.check(dfe.lib().dfe_min_dim0_f32(ctx.handle, out.data_ptr(), 256, M, mn.data_ptr(), mi.data_ptr()))
    view = out.cpu().numpy().reshape(256, M)
    assert np.array_equal(mn.cpu().numpy(), view.min(axis=0)) and np.array_equal(mi.cpu().numpy(), view.argmin(axis=0) + 1)



@pytest.mark.parametrize("n,M", [(256, 1000), (37, 129), (5, 64), (1, 70), (16, 63), (100, 4099)])
def test_min_dim0_is_the_sequential_first_minimum(dfe, cuda, n, M):
    """dfe_min_dim0_f32 (tests/time_matching.lua:41-43, `output:min(1)`): the row the loop `best = in[0]; if (v < best) ...` ends on -- ties
    go to the first row, +inf never replaces, a NaN in row 0 stays and a NaN elsewhere never wins -- for row counts that do not divide
    into the kernel's 16 row groups and column counts that do not fill its 64-column blocks."""
    rng = np.random.default_rng(n * 1000 + M)
    a = rng.integers(0, 6, size=(n, M)).astype(np.float32)            # many ties
    a[:, 3] = np.inf
    if n > 2:
        a[2, 5] = np.nan
        a[0, 7] = np.nan
        a[n - 1, 9] = -1.0
    ctx = dfe.get_ctx(0)
    t = torch.from_numpy(a).to(cuda)
    mn = torch.empty((M,), device=cuda)
    mi = torch.empty((M,), dtype=torch.int64, device=cuda)
    ctx.check(dfe.lib().dfe_min_dim0_f32(ctx.handle, t.data_ptr(), n, M, mn.data_ptr(), mi.data_ptr()))
    best, bi = a[0].copy(), np.zeros(M, dtype=np.int64)
    with np.errstate(invalid="ignore"):
        for r in range(1, n):
            w = a[r] < best
            best[w] = a[r][w]
            bi[w] = r
    assert np.array_equal(mn.cpu().numpy(), best, equal_nan=True) and np.array_equal(mi.cpu().numpy(), bi + 1)


def test_pipelined_ingest_equals_the_serial_u8_entry(dfe, cuda):
    """dfe_ingest_submit_u8 + dfe_flow_depth_pair_u8_slot (the upload of pair i+1 on the ctx's copy stream beside the step of pair i, two
    device slots, event-ordered): a stream of DIFFERENT pairs through the pipeline gives, pair by pair, exactly what dfe_flow_depth_pair_u8
    gives on the same frames -- slots are not overwritten before they are consumed, results do not arrive early."""
    from ctypes import c_int

    ctx = dfe.get_ctx(0)
    lib = dfe.lib()
    H, W, k, win = 96, 140, 7, 17
    rng = np.random.default_rng(31)
    pairs = []
    for i in range(5):
        u0 = rng.integers(0, 256, (3, H, W), dtype=np.uint8)
        u1 = np.roll(u0, (i % 3 - 1, 2 - i % 4), axis=(1, 2))
        pairs.append((torch.from_numpy(u0).pin_memory(), torch.from_numpy(u1).pin_memory()))

    def outs():
        return torch.full((2, H, W), -9.0, device=cuda), *(torch.full((H, W), -9.0, device=cuda) for _ in range(3))

    want = []
    for h0, h1 in pairs:
        o = outs()
        d0, d1 = h0.to(cuda), h1.to(cuda)
        ctx.check(lib.dfe_flow_depth_pair_u8(ctx.handle, d0.data_ptr(), d1.data_ptr(), 3, H, W, k, win, win, 70.0, 40.0, 0.21, 1.0, *(t.data_ptr() for t in o)))
        want.append(o)
    torch.cuda.synchronize()
    got = [outs() for _ in pairs]
    slot = c_int()
    ctx.check(lib.dfe_ingest_submit_u8(ctx.handle, pairs[0][0].data_ptr(), pairs[0][1].data_ptr(), 3 * H * W, byref(slot)))
    cur = slot.value
    for i in range(len(pairs)):
        if i + 1 < len(pairs):
            ctx.check(lib.dfe_ingest_submit_u8(ctx.handle, pairs[i + 1][0].data_ptr(), pairs[i + 1][1].data_ptr(), 3 * H * W, byref(slot)))
            nxt = slot.value
            assert nxt != cur
        ctx.check(lib.dfe_flow_depth_pair_u8_slot(ctx.handle, cur, 3, H, W, k, win, win, 70.0, 40.0, 0.21, 1.0, *(t.data_ptr() for t in got[i])))
        cur = nxt
    torch.cuda.synchronize()
    for i, (g, w_) in enumerate(zip(got, want)):
        for a, b in zip(g, w_):
            assert torch.equal(a, b), i
    assert lib.dfe_flow_depth_pair_u8_slot(ctx.handle, 7, 3, H, W, k, win, win, 70.0, 40.0, 0.21, 1.0, *(t.data_ptr() for t in got[0])) != 0   # no such slot


def test_device_alloc_and_arena_placement_do_not_change_results(dfe, cuda):
    """dfe_device_alloc / dfe_device_free (include/dfe.h): a caller-owned volume in the library's memory (physically contiguous where the driver
    grants it) holds exactly what a torch-allocated one holds and what the oracle computes; the one-call pipeline gives the same bits with the
    ctx arena contiguous (default) or a plain hipMalloc (option arena_contig = 0, a fresh context each so that each really allocates)."""
    from ctypes import c_int, c_void_p
    from depth_estimation_amd.context import Context

    lib = dfe.lib()
    ctx = dfe.get_ctx(0)
    H, W, k, win = 70, 150, 7, 33
    f0, f1, _, (cx, cy) = rp.synth_pair(H, W, C=3, seed=12, max_flow=9)
    t0, t1 = T(f0, cuda), T(f1, cuda)
    Ho, Wo = H - k + 1 - win + 1, W - k + 1 - win + 1
    ref = torch.empty((Ho, Wo, win, win), device=cuda)
    ctx.check(lib.dfe_ssd_cost_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, k, k, win, win, ref.data_ptr()))
    assert np.array_equal(ref.cpu().numpy(), orc.ssd_cost_volume(f0, f1, k, k, win, win))
    p, contig = c_void_p(), c_int(-1)
    ctx.check(lib.dfe_device_alloc(ctx.handle, ref.numel() * 4, byref(p), byref(contig)))
    assert p.value and contig.value in (0, 1)
    ctx.check(lib.dfe_ssd_cost_volume_f32(ctx.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, k, k, win, win, p))
    back = torch.empty_like(ref)
    torch.cuda.synchronize()
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(c_void_p(back.data_ptr()), p, ctypes.c_size_t(ref.numel() * 4), c_int(3)) == 0
    assert torch.equal(back, ref)
    ctx.check(lib.dfe_device_free(ctx.handle, p))
    ctx.check(lib.dfe_device_free(ctx.handle, None))
    assert lib.dfe_device_alloc(ctx.handle, 0, byref(p), None) != 0
    res = []
    for contig_opt in (1, 0):
        c2 = Context(0)
        c2.set_option("arena_contig", contig_opt)
        o = [torch.full((2, H, W), -9.0, device=cuda)] + [torch.full((H, W), -9.0, device=cuda) for _ in range(3)]
        c2.check(lib.dfe_flow_depth_pair_f32(c2.handle, t0.data_ptr(), t1.data_ptr(), 3, H, W, k, win, win, cx, cy, 0.21, *(t.data_ptr() for t in o)))
        torch.cuda.synchronize()
        res.append(o)
        del c2
    for a, b in zip(*res):
        assert torch.equal(a, b)
