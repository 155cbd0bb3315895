"""CPU suite (-m "not gpu"): pins the oracle against the reference's own data-free known-answer
tests and the committed golden fixtures.  No GPU, no /root/reference access."""
import json
import math
import os

import numpy as np
import pytest

from tests import oracle as orc
from tests import refpath as rp

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# --- cartesian_groundtruth_cc_testme (radial/radial_opticalflow_groundtruth.lua:170-210) ----------
@pytest.mark.parametrize("case", [0, 1, 2, 3])
@pytest.mark.parametrize("via_unfold", [False, True])
def test_kat_planted_flow_recovered(case, via_unfold):
    im1, im2, fb, (hK, wK, hW, wW) = rp.kat_case(case)
    res = rp.dense_flow_oracle(im1, im2, hW, wW, hK, wK, via_unfold=via_unfold)
    flow = res["flowp"]
    diff = (fb - flow[:2]) * flow[2]
    assert flow.shape == (4,) + im1.shape[1:]
    assert flow[2].sum() > 0
    assert np.abs(diff).sum() == 0  # :186-192


def test_unfold_matches_fused_volume_bitwise():
    # A0+A1 fused restatement == the literal unfold -> crop -> SpatialMatching composition
    im1, im2, _, (hK, wK, hW, wW) = rp.kat_case(3)
    a = rp.dense_flow_oracle(im1, im2, hW, wW, hK, wK, via_unfold=False)["cost"]
    b = rp.dense_flow_oracle(im1, im2, hW, wW, hK, wK, via_unfold=True)["cost"]
    assert np.array_equal(a, b)


def test_even_window_asymmetric_range():
    # hWin=12 -> displacements -5..+6; decode of first / last class
    y, x = orc.x2yx(np.array([1, 12 * 15], np.int64), 12, 15)
    assert (y[0], x[0]) == (-5, -7) and (y[1], x[1]) == (6, 7)


# --- SpatialMatching == brute-force SSD (tests/test_multiscale.lua:135-166) ----------------------
def test_matching_equals_bruteforce_ssd():
    rng = np.random.default_rng(3)
    K, H1, W1, mh, mw = 5, 6, 7, 4, 3
    in1 = rng.standard_normal((K, H1, W1)).astype(np.float32)
    in2 = rng.standard_normal((K, H1 + mh - 1, W1 + mw - 1)).astype(np.float32)
    out = orc.spatial_matching(in1, in2, mh, mw)
    for y in range(H1):
        for x in range(W1):
            best, ib, jb = 1e25, 0, 0
            for i in range(mh):
                for j in range(mw):
                    s = float(((in1[:, y, x].astype(np.float64) - in2[:, y + i, x + j]) ** 2).sum())
                    assert abs(out[y, x, i, j] - s) <= 1e-5 * max(1.0, s)
                    if s < best:
                        best, ib, jb = s, i, j
            m = int(out[y, x].reshape(-1).argmin())
            assert (m // mw, m % mw) == (ib, jb)  # :161-165


def test_radial_matching_is_vertical_only():
    rng = np.random.default_rng(4)
    K, H1, W, hw = 4, 9, 11, 5
    in1 = rng.standard_normal((K, H1, W)).astype(np.float32)
    in2 = rng.standard_normal((K, H1 + hw - 1, W)).astype(np.float32)
    out = orc.radial_matching(in1, in2, hw)
    ref = np.stack([((in1 - in2[:, d : d + H1]) ** 2).sum(0) for d in range(hw)], -1)
    assert np.allclose(out, ref, rtol=1e-5, atol=1e-5)


# --- centre tie-break (radial/radial_opticalflow_groundtruth.lua:88-94) -------------------------
def test_argbest_center_tie_break_and_first_wins():
    N = 9
    vol = np.array([[3, 1, 5, 1, 1, 7, 1, 9, 9],      # min 1 at cells 2,4,5,7 (1-based); centre=5 ties -> 5
                    [3, 1, 5, 1, 2, 7, 1, 9, 9],      # centre (2) is not the min -> first min = 2
                    [0, 0, 0, 0, 0, 0, 0, 0, 0]], np.float32)
    idx, best = orc.argbest_center(vol, 5, take_max=False)
    assert idx.tolist() == [5, 2, 5] and best.tolist() == [1, 1, 0]
    idx, _ = orc.argbest_center(vol, 5, take_max=True)
    assert idx.tolist() == [8, 8, 5]
    idx, _ = orc.argbest_center(vol, 0, take_max=False)  # override disabled
    assert idx.tolist() == [2, 2, 1]


# --- extractOutput: hand-derived from extract_output.cpp:63-155 ---------------------------------
def test_extract_output_hand_vector():
    v = np.array([[0.05, 0.15, 0.12, 0.05, 0.9, 0.25, 0.2, 0.225, 0.3],   # thr .11 -> M=8: keeps .15 .12 .9 .25 .2 .225 .3
                  [0.0, 0.05, 0.1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]], np.float32)  # nothing above -> untouched
    imaxs = np.full(2, -7, np.int64)
    scores = np.full(2, -3.0, np.float32)
    orc.extract_output(v, 0.11, imaxs, scores)
    srt = sorted([0.15, 0.12, 0.9, 0.25, 0.2, 0.225, 0.3], reverse=True) + [0.0]
    expect = sum(np.cumsum(np.array(srt, np.float32), dtype=np.float32).astype(np.float64))
    assert imaxs.tolist() == [5, -7]
    assert scores[1] == -3.0
    assert abs(scores[0] - expect) < 1e-6
    # thr >= .2 -> M=4, scan stops after the 4th hit (.9 .25 .225 .3 in index order; .2 is not > .2f? it is > 0.2 as float)
    imaxs[:] = -7
    scores[:] = -3
    orc.extract_output(v, 0.21, imaxs, scores)
    kept = [0.9, 0.25, 0.225, 0.3]
    srt = sorted(kept, reverse=True)
    expect = sum(np.cumsum(np.array(srt, np.float32), dtype=np.float32).astype(np.float64))
    assert imaxs[0] == 5 and abs(scores[0] - expect) < 1e-6


def test_extract_output_tie_order_follows_network():
    # equal keys never swap (sortswap uses strict >): with 4 equal values the index that ends up
    # first is the one the fixed network leaves in slot 0, i.e. the first kept.
    v = np.array([[0.5, 0.5, 0.5, 0.5, 0.1]], np.float32)
    imaxs = np.zeros(1, np.int64)
    scores = np.zeros(1, np.float32)
    orc.extract_output(v, 0.3, imaxs, scores)
    assert imaxs[0] == 1 and abs(scores[0] - (0.5 + 1.0 + 1.5 + 2.0)) < 1e-6
    v = np.array([[0.4, 0.5, 0.5, 0.45, 0.1]], np.float32)   # network: (0,2)(1,3)(0,1)(2,3)(1,2)
    orc.extract_output(v, 0.3, imaxs, scores)
    assert imaxs[0] == 3  # after (0,2) swap slot0 holds index 3 (value .5); (0,1): .5 > .5 false -> stays


def test_extract_marginalized():
    v = np.array([[0.3, 0.3, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0]], np.float32)
    ret = np.full(2, -1, np.int64)
    gd = np.full(2, 5, np.int64)
    orc.extract_output_marginalized(v, 0.25, 0.8, ret, gd)
    assert ret.tolist() == [1, -1] and gd.tolist() == [1, 0]   # acc = .3+.6+.6+.6 = 2.1 >= .8


# --- multiscale codec (tests/test_multiscale.lua:57-80) -----------------------------------------
GEOS = [(8, 8, [1, 2]), (8, 8, [1, 2, 4]), (4, 4, [1, 2, 4, 8]), (16, 16, [1, 2, 4, 8]), (8, 8, [1])]


@pytest.mark.parametrize("maxh,maxw,ratios", GEOS)
def test_codec_round_trip(maxh, maxw, ratios):
    mh, mw = maxh * ratios[-1], maxw * ratios[-1]
    for i in range(-math.ceil(mh / 2) + 1, mh // 2 + 1):
        for j in range(-math.ceil(mw / 2) + 1, mw // 2 + 1):
            rc, y, x = orc.x2yx_multi_number(maxh, maxw, ratios, orc.yx2x_multi(maxh, maxw, ratios, i, j))
            assert rc == 0
            tol = None
            for r in ratios:
                if abs(i) < maxh * r and abs(j) < maxw * r:
                    tol = r
            assert abs(y - i) < tol and abs(x - j) < tol   # :61-71
    maxx = maxh * maxw
    for k in range(1, len(ratios)):
        maxx += maxh * maxw * (1 - (ratios[k - 1] / ratios[k]) ** 2)
    assert orc.multi_nclasses(maxh, maxw, ratios) == int(maxx)
    for i in range(1, int(maxx) + 1):
        rc, y, x = orc.x2yx_multi_number(maxh, maxw, ratios, i)
        assert rc == 0 and orc.yx2x_multi(maxh, maxw, ratios, y, x) == i   # :78-80
    assert orc.x2yx_multi_number(maxh, maxw, ratios, int(maxx) + 1)[0] != 0
    assert orc.x2yx_multi_number(maxh, maxw, ratios, 0)[0] != 0


def test_codec_known_values_and_compat():
    # (8,8,{1,2,4}): 160 classes, middleIndex 28, id 65 -> (-6,-6), reach -12..+16
    assert orc.multi_nclasses(8, 8, [1, 2, 4]) == 160
    assert orc.yx2x_multi(8, 8, [1, 2, 4], 0, 0) == 28
    assert orc.x2yx_multi_number(8, 8, [1, 2, 4], 65)[1:] == (-6, -6)
    assert orc.x2yx_multi_number(8, 8, [1, 2, 4], 113)[1:] == (-12, -12)
    assert orc.x2yx_multi_number(8, 8, [1, 2, 4], 160)[1:] == (16, 16)
    # the shipped vectorised C body diverges (x2yxMulti2.c): id 65 -> (-3,-3), ids 151..160 never written
    ids = np.arange(1, 161, dtype=np.int64)
    y, x = orc.x2yx_multi_compat_c(8, 8, [1, 2, 4], ids, fill=-999)
    assert (y[64], x[64]) == (-3, -3)
    assert (y[150:] == -999).all() and (x[150:] == -999).all()


# --- cascade (CascadingAddTable.lua:108-135, tests/test_multiscale.lua:169-193 with HEAD's sum) --
def test_cascading_add_matches_definition():
    rng = np.random.default_rng(7)
    ratios, mh, mw, P = [1, 2, 4], 8, 8, 5
    ins = [rng.random((P, mh, mw), dtype=np.float32) for _ in ratios]
    rc, outs = orc.cascading_add(ins, ratios, mh, mw)
    assert rc == 0
    cy, cx = math.ceil(mh / 2), math.ceil(mw / 2)
    assert np.array_equal(outs[-1], ins[-1])
    for i in range(len(ratios)):
        s = np.zeros((P, mh, mw), np.float64)
        for ii in range(-cy + 1, cy + 1):
            for jj in range(-cx + 1, cx + 1):
                for j in range(i, len(ratios)):
                    r = ratios[j] // ratios[i]
                    s[:, ii + cy - 1, jj + cx - 1] += ins[j][:, math.ceil(ii / r) + cy - 1, math.ceil(jj / r) + cx - 1]
        assert np.allclose(outs[i], s, atol=1e-5)   # :176-186 (sum, not mean: HEAD)
    rc, _ = orc.cascading_add([np.zeros((1, 2, 2), np.float32)] * 2, [1, 2], 2, 2)
    assert rc != 0   # CascadingAddTable.lua:121-124 incompatibility error


def test_ring_layout_matches_codec():
    # class k of the ring-extracted vector must be the cascade cell that x2yxMultiNumber(k) decodes to
    ratios, mh, mw, H, W = [1, 2, 4], 8, 8, 4, 8
    probs = []
    for s, r in enumerate(ratios):
        a = np.zeros((H // r, W // r, mh, mw), np.float32)
        for ty in range(mh):
            for tx in range(mw):
                a[:, :, ty, tx] = 1000 * (s + 1) + 10 * ty + tx   # encodes (scale, cell)
        probs.append(a)
    # disable accumulation effects: compare on the coarsest-only contribution by zeroing finer scales' adds
    rc, out = orc.cascade_ring(probs, ratios, H, W, mh, mw)
    assert rc == 0 and out.shape == (H, W, 160)
    rc2, casc = orc.cascading_add([np.repeat(np.repeat(p, r, 0), r, 1).reshape(-1, mh, mw) for p, r in zip(probs, ratios)], ratios, mh, mw)
    for k in range(1, 161):
        _, y, x = orc.x2yx_multi_number(mh, mw, ratios, k)
        s = 0 if k <= 64 else (1 if k <= 112 else 2)
        ty, tx = y // ratios[s] + math.ceil(mh / 2) - 1, x // ratios[s] + math.ceil(mw / 2) - 1
        assert out[0, 0, k - 1] == casc[s][0, ty, tx]


# --- flow -> depth ------------------------------------------------------------------------------
def test_flow_to_depth_cartesian_quirk():
    H, W = 4, 6
    flow = np.zeros((2, H, W), np.float32)
    flow[0] = 1.0   # dy
    flow[1] = 0.5   # dx
    d, c = orc.flow_to_depth_cartesian(flow, W / 2, H / 2)
    dn = math.sqrt(1.25)
    assert abs(d[0, 0] - min(math.sqrt(9 + 4) / dn, W / 2)) < 1e-5
    assert c[0, 0] == (1.0 if (-3 * 0.5 + 1.0) > 0.125 else 0.0)   # px*dx + dy*dy (sic), test_opticalflow.lua:181
    d2, c2 = orc.flow_to_depth_cartesian(np.zeros((2, H, W), np.float32), W / 2, H / 2)
    assert (d2 == W / 2).all() and (c2 == 1).all()


# --- golden fixtures ----------------------------------------------------------------------------
def test_golden_fixtures_reproduce():
    path = os.path.join(GOLD, "golden_v1.npz")
    assert os.path.exists(path), "run tests/golden/make_golden.py"
    g = np.load(path)
    im1, im2, _, (hK, wK, hW, wW) = rp.kat_case(3, seed=5, C=3, h=24, w=28)
    assert np.array_equal(g["kat3_im1"], im1)
    res = rp.dense_flow_oracle(im1, im2, hW, wW, hK, wK)
    for k in ("cost", "idx", "scores", "imaxs"):
        assert np.array_equal(g["kat3_" + k], res[k]), k
    ids = np.arange(1, 161, dtype=np.int64)
    _, y, x = orc.x2yx_multi(8, 8, [1, 2, 4], ids)
    assert np.array_equal(g["codec_8_8_124_y"], y) and np.array_equal(g["codec_8_8_124_x"], x)
    yc, xc = orc.x2yx_multi_compat_c(8, 8, [1, 2, 4], ids, fill=-999)
    assert np.array_equal(g["codec_compat_y"], yc) and np.array_equal(g["codec_compat_x"], xc)


# --- A16 / A17 / A18 hand-checked -----------------------------------------------------------------
def test_postprocess_mode_and_median_hand_checked():
    H, W, k = 6, 7, 3
    flow = np.zeros((2, H, W), np.float32)
    flow[0] = 2.0          # y
    flow[1] = -1.0         # x
    flow[0, 2, 2] = 5.4    # one outlier inside every 3x3 window around it
    flow[1, 2, 2] = 3.6
    mask = np.ones((H, W), np.float32)
    rc, out = orc.postprocess_image(flow, mask, k, "max")
    assert rc == 0
    # windows are anchored at i < H-k, j < W-k and written at (i+1, j+1): rows 1..3, cols 1..4 (as shipped)
    assert (out[0, 1:4, 1:5] == 2.0).all() and (out[1, 1:4, 1:5] == -1.0).all()
    assert (out[:, 0, :] == -1.0).all() and (out[:, 4:, :] == -1.0).all()   # untouched border = 0 + m, m = min = -1
    rc, med = orc.postprocess_image(flow, mask, k, "med")
    assert rc == 0 and (med[0, 1:4, 1:5] == 2.0).all() and (med[1, 1:4, 1:5] == -1.0).all() and (med[:, 0, :] == 0).all()
    mask[:] = 0
    rc, med = orc.postprocess_image(flow, mask, k, "med")
    assert (med == 0).all()                                                  # n = 0 -> tmp[0] of the zeroed buffer
    assert orc.postprocess_image(flow, np.ones((H, W), np.float32), 7, "med")[0] != 0   # 49 values > the 32-value buffer
    flow[0, 0, 0] = 40
    assert orc.postprocess_image(flow, np.ones((H, W), np.float32), k, "max")[0] != 0   # outside the 16x16 histogram


def test_enlarge_mask_hand_checked():
    m = np.zeros((5, 8), np.float32)
    m[1:4, 1:7] = 1
    out = orc.enlarge_mask(m, 2, 1)
    exp = np.zeros((5, 8), np.float32)
    exp[2, 3:5] = 1      # rows: 2 px off each side of cols 1..6 -> 3..4; columns: 1 px off top and bottom of rows 1..3 -> 2
    assert np.array_equal(out, exp)


def test_output_extractor_hand_checked():
    p = np.zeros((1, 6), np.float32)   # 2 x 3 window, all mass on cell (i=2, j=3)
    p[0, 5] = 1.0
    x, y = orc.output_extractor(p, 2, 3)
    assert x[0] == 3.0 and y[0] == 2.0
    p[:] = 1 / 6
    x, y = orc.output_extractor(p, 2, 3)
    assert abs(x[0] - 2.0) < 1e-6 and abs(y[0] - 1.5) < 1e-6


@pytest.mark.parametrize("ratios,mh,mw", [([1, 2, 4], 8, 8), ([1, 2], 8, 16), ([1, 2, 4, 8], 16, 16), ([1], 8, 8), ([1, 4], 8, 8)])
def test_cascading_add_backward_is_the_adjoint_of_forward(ratios, mh, mw):
    """A4b, pinned the way the reference pins it (tests/test_cascad.lua:22 nn.Jacobian.testJacobian): the cascade is
    linear, so <J x, g> == <x, J^T g> for random x, g, and J^T g equals the finite-difference Jacobian column sums."""
    rng = np.random.default_rng(len(ratios) * 100 + mh)
    P = 3
    xs = [rng.standard_normal((P, mh, mw)).astype(np.float32) for _ in ratios]
    gs = [rng.standard_normal((P, mh, mw)).astype(np.float32) for _ in ratios]
    rc, ys = orc.cascading_add(xs, ratios, mh, mw)
    rc2, gis = orc.cascading_add_backward(gs, ratios, mh, mw)
    assert rc == 0 and rc2 == 0
    lhs = sum(float((y.astype(np.float64) * g).sum()) for y, g in zip(ys, gs))
    rhs = sum(float((x.astype(np.float64) * gi).sum()) for x, gi in zip(xs, gis))
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))
    # explicit Jacobian on one pixel: d(sum_s <y_s, g_s>)/d x_s[n] by central differences (exact for a linear map)
    for s_i in range(len(ratios)):
        for n in (0, mw + 1, mh * mw - 1):
            e = [np.zeros((1, mh, mw), np.float32) for _ in ratios]
            e[s_i].reshape(-1)[n] = 1.0
            _, ye = orc.cascading_add(e, ratios, mh, mw)
            col = sum(float((y[0].astype(np.float64) * g[0]).sum()) for y, g in zip(ye, gs))
            assert abs(col - float(gis[s_i][0].reshape(-1)[n])) <= 1e-4 * max(1.0, abs(col))
    # incompatible geometry is refused like the forward
    rc3, _ = orc.cascading_add_backward([np.zeros((1, 2, 2), np.float32)] * 2, [1, 2], 2, 2)
    assert rc3 != 0


def test_flow_to_depth_ardrone_hand_vector():
    """A12(iii) ardrone/ardrone_api.cpp:99-140 on a vector worked by hand: the window is [i-3, i+3) (half-open), masked
    samples only, first maximum of the histogram wins, |mode| < 1.1 -> 100, the centre column and unmasked pixels get
    conf 0."""
    H, W = 5, 8
    xflow = np.zeros((H, W), np.float32)
    xflow[:, :4] = 2.4      # rounds to 2
    xflow[:, 4:] = -3.6     # rounds to -4
    mask = np.ones((H, W), np.float32)
    mask[0, 0] = 0.0        # not a sample, no output
    mask[1, 1] = 0.3        # a sample (non-zero) but conf 0 (<= 0.5)
    d, c = orc.flow_to_depth_ardrone(xflow, mask, 0.5)
    middlex = W // 2
    assert c[0, 0] == 0 and d[0, 0] == 0 and c[1, 1] == 0
    assert np.all(c[:, middlex] == 0)
    # pixel (2, 1): window columns [-2..3] -> 0..3 all "2": mode 2 -> 0.5*|1-4|/2
    assert c[2, 1] == 1 and d[2, 1] == np.float32(0.5 * 3 / 2)
    # pixel (2, 6): window columns 3..7: one column of 2 (col 3), four of -4 -> mode -4 -> 0.5*2/4
    assert d[2, 6] == np.float32(0.5 * 2 / 4)
    # pixel (2, 5): columns 2..7 -> two columns of 2, four of -4 -> -4
    assert d[2, 5] == np.float32(0.5 * 1 / 4)
    # tie: columns 1..6 around i=4 is the centre column (conf 0); i=3: columns 0..5 -> four columns of 2, two of -4 -> 2
    assert d[2, 3] == np.float32(0.5 * 1 / 2)
    # small flow -> 100
    d2, c2 = orc.flow_to_depth_ardrone(np.full((H, W), 0.4, np.float32), np.ones((H, W), np.float32), 1.0)
    assert np.all(d2[:, [0, 1, 2, 3, 5, 6, 7]] == 100.0) and np.all(c2[:, middlex] == 0)
    # out-of-range samples (the reference's unchecked histogram index) are skipped
    d3, _ = orc.flow_to_depth_ardrone(np.full((H, W), 40.0, np.float32), np.ones((H, W), np.float32), 1.0)
    assert np.all(d3[:, 0] == 100.0)


def test_matcher_gradients_are_the_jacobian_of_the_forward():
    """N2: orc_spatial_matching_backward / orc_radial_matching_backward against central differences of the forward
    restatements (quadratic in the inputs, so the difference quotient is exact on small integers) -- the reference
    holds no test for these un-vendored modules; method of tests/test_cascad.lua:22."""
    rng = np.random.default_rng(11)
    K, H1, W1, mh, mw = 3, 5, 6, 3, 4
    in1 = rng.integers(-4, 5, (K, H1, W1)).astype(np.float32)
    in2 = rng.integers(-4, 5, (K, H1 + mh - 1, W1 + mw - 1)).astype(np.float32)
    go = rng.integers(-3, 4, (H1, W1, mh, mw)).astype(np.float32)
    g1, g2 = orc.spatial_matching_backward(in1, in2, go, mh, mw)
    def loss(a, b):
        return float((orc.spatial_matching(a, b, mh, mw).astype(np.float64) * go).sum())
    for arr, g, which in ((in1, g1, 0), (in2, g2, 1)):
        for idx in [(0, 0, 0), (1, 2, 3), (K - 1, arr.shape[1] - 1, arr.shape[2] - 1), (2, 1, 0)]:
            p, m = arr.copy(), arr.copy()
            p[idx] += 1.0
            m[idx] -= 1.0
            fd = (loss(p, in2) - loss(m, in2)) / 2.0 if which == 0 else (loss(in1, p) - loss(in1, m)) / 2.0
            assert fd == float(g[idx]), (which, idx, fd, g[idx])
    hW = 4
    r1 = rng.integers(-4, 5, (K, H1, W1)).astype(np.float32)
    r2 = rng.integers(-4, 5, (K, H1 + hW - 1, W1)).astype(np.float32)
    rgo = rng.integers(-3, 4, (H1, W1, hW)).astype(np.float32)
    h1, h2 = orc.radial_matching_backward(r1, r2, rgo, hW)
    # the radial matcher is the 2-D one with a 1-column window
    e1, e2 = orc.spatial_matching_backward(r1, r2, rgo.reshape(H1, W1, hW, 1), hW, 1)
    assert np.array_equal(h1, e1) and np.array_equal(h2, e2)
    def rloss(a, b):
        return float((orc.radial_matching(a, b, hW).astype(np.float64) * rgo).sum())
    p, m = r2.copy(), r2.copy()
    p[1, 3, 2] += 1.0
    m[1, 3, 2] -= 1.0
    assert (rloss(r1, p) - rloss(r1, m)) / 2.0 == float(h2[1, 3, 2])


# --- the reference's remaining data-free checks, restated on the oracle ---------------------------
@pytest.mark.parametrize("maxh", [8, 7])
def test_pyramid_windows_are_centred_on_the_pixel(maxh):
    """tests/test_multiscale.lua:111-133: at every pyramid scale the window is centred on the pixel ("centered on ceil-1
    AFTER the matching"): matching a frame against itself costs exactly 0 at cell (ceil(maxh/2), ceil(maxw/2)) (1-based)
    and nowhere else; and content that moved by (2r, -r) pixels is found two cells below / one cell left of the centre at
    the scale of ratio r."""
    rng = np.random.default_rng(maxh)
    H, W, k = 64, 80, 7
    I0 = rng.random((3, H, W), dtype=np.float32)
    c = math.ceil(maxh / 2) - 1
    for r in (1, 2, 4):
        v = orc.pyramid_scale_volume(I0, I0, r, k, k, maxh, maxh)
        assert v.shape == (H // r, W // r, maxh, maxh)
        assert (v[:, :, c, c] == 0).all()
        inner = v[8 // r + 2 : -(8 // r) - 2, 8 // r + 2 : -(8 // r) - 2].reshape(-1, maxh * maxh)
        others = np.delete(inner, c * maxh + c, axis=1)
        assert (others > 0).all()
        I1 = np.zeros_like(I0)
        I1[:, 2 * r :, : W - r] = I0[:, : H - 2 * r, r:]          # I1(p + (2r, -r)) = I0(p)
        v = orc.pyramid_scale_volume(I0, I1, r, k, k, maxh, maxh)
        m = 16 // r + 2
        am = v[m:-m, m:-m].reshape(v.shape[0] - 2 * m, v.shape[1] - 2 * m, -1).argmin(-1)
        assert (am == (c + 2) * maxh + (c - 1)).all()
        assert (v[m:-m, m:-m, c + 2, c - 1] == 0).all()


@pytest.mark.parametrize("maxh,ratios", [(8, [1, 2, 4]), (8, [1, 2]), (16, [1, 2, 4, 8]), (4, [1, 2, 4, 8])])
def test_ring_extraction_block_layout(maxh, ratios):
    """tests/test_multiscale.lua:195-214 ("check complex reshaping"), restated: scale 1 leaves all its cells; for scale
    i >= 2 the extracted vector, cut into the four blocks top (dh x maxw), left (lih x dw), right (lih x dw), bottom
    (dh x maxw) and put back in place, is the cascade output of that scale with its middle lih x liw cells zeroed."""
    rng = np.random.default_rng(maxh + len(ratios))
    maxw, H, W = maxh, ratios[-1], 2 * ratios[-1]
    probs = [rng.random((H // r, W // r, maxh, maxw), dtype=np.float32) for r in ratios]
    rc, out = orc.cascade_ring(probs, ratios, H, W, maxh, maxw)
    assert rc == 0
    up = [np.repeat(np.repeat(p, r, 0), r, 1).reshape(-1, maxh, maxw) for p, r in zip(probs, ratios)]
    rc, casc = orc.cascading_add(up, ratios, maxh, maxw)
    assert rc == 0
    out = out.reshape(H * W, -1)
    assert np.array_equal(out[:, : maxh * maxw], casc[0].reshape(H * W, -1))           # reshaper.modules[1].output == cascad_out[1]
    off = maxh * maxw
    for i in range(1, len(ratios)):
        liw = round(maxw * ratios[i - 1] / ratios[i]); dw = round((maxw - liw) / 2)
        lih = round(maxh * ratios[i - 1] / ratios[i]); dh = round((maxh - lih) / 2)
        blockt = casc[i].copy()
        blockt[:, dh : dh + lih, dw : dw + liw] = 0
        n = 2 * dh * maxw + 2 * lih * dw
        reshaped = out[:, off : off + n]
        block = np.zeros_like(blockt)
        block[:, :dh] = reshaped[:, : dh * maxw].reshape(-1, dh, maxw)
        block[:, dh : dh + lih, :dw] = reshaped[:, dh * maxw : dh * maxw + lih * dw].reshape(-1, lih, dw)
        block[:, dh : dh + lih, dw + lih : dw + lih + dw] = reshaped[:, dh * maxw + lih * dw : dh * maxw + 2 * lih * dw].reshape(-1, lih, dw)
        block[:, dh + lih : dh + lih + dh] = reshaped[:, dh * maxw + 2 * lih * dw : n].reshape(-1, dh, maxw)
        assert np.array_equal(block, blockt)
        off += n
    assert off == out.shape[1]


def test_patch_mode_matching_argmax_is_block_matching_ground_truth():
    """tests/test_patches.lua:46-60: one 16x16 patch of frame 1 (K = 3*16*16 features at a 1x1 map) against the 16x16
    candidate patches of a (16+16-1)^2 region of frame 2, through unfold + nn.SpatialMatching(16, 16): the arg-max of
    -output is the class of the true displacement (here: planted; the reference takes it from its ground-truth files)."""
    rng = np.random.default_rng(3)
    k, maxh = 16, 16
    I0 = rng.random((3, 80, 90), dtype=np.float32)
    c = math.ceil(maxh / 2) - 1                                    # 0-based centre cell: displacement 0
    for fy, fx in ((0, 0), (3, -5), (-7, 8), (8, -7), (-2, 2)):
        I1 = np.roll(I0, (fy, fx), axis=(1, 2))                    # I1(p + f) = I0(p)
        y, x = 30, 35
        in1 = orc.unfold(I0[:, y : y + k, x : x + k], k, k)        # K x 1 x 1
        reg = I1[:, y - c : y - c + maxh + k - 1, x - c : x - c + maxh + k - 1]
        in2 = orc.unfold(reg, k, k)                                # K x 16 x 16
        assert in1.shape == (3 * k * k, 1, 1) and in2.shape == (3 * k * k, maxh, maxh)
        out = orc.spatial_matching(in1, in2, maxh, maxh)           # 1 x 1 x 16 x 16
        m = int((-out).reshape(-1).argmax()) + 1                   # 1-based class, as torch's max
        target = (fy + math.ceil(maxh / 2) - 1) * maxh + fx + math.ceil(maxh / 2)   # yx2x(centered2onebased(fy, fx)), opticalflow_model.lua:12-34
        assert m == target and out.reshape(-1)[m - 1] == 0
