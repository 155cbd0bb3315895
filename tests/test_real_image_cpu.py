"""CPU suite, BASELINE configs[0]: the celiu car1 -> car2 pair (tests/golden/celiu/*.jpg are byte copies of
/root/reference/celiu/car1.jpg, car2.jpg and celiu/output/car_flow.jpg -- data), halved to 320x240 as celiu/demoflow.m:11-12 does,
through the single-scale dense path on the CPU oracle: 7x7 patch, +-8 search (17x17), i.e. the tests/time_matching.lua /
compute_cartesian_groundtruth_cross_correlation path (unfold -> SpatialMatching -> min + centre tie-break -> decode ->
extractOutput -> pad back).  Plumbing: shapes, ranges, the unfold route == the fused route; and agreement with the variational
flow of the named ground truth -- Ce Liu's Coarse2FineTwoFrames on this pair -- as far as the reference HOLDS it:
celiu/output/car_flow.jpg, the flow colour-coded by celiu/flowToColor.m / computeColor.m, decoded by tests/flowcolor.py into the
flow's direction and its magnitude relative to the field's maximum (the absolute scale is printed by flowToColor, not stored)."""
import os

import numpy as np
import pytest

from tests import oracle as orc
from tests import refpath as rp

CELIU = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "celiu")


def load_pair(size):
    from PIL import Image

    out = []
    for n in ("car1.jpg", "car2.jpg"):
        im = Image.open(os.path.join(CELIU, n)).convert("RGB")
        assert im.size == (640, 480)
        if size != (640, 480):
            im = im.resize(size, Image.BICUBIC)
        out.append(np.ascontiguousarray(np.asarray(im, np.float32).transpose(2, 0, 1)))   # C x H x W, uint8 values in fp32
    return out


def test_car_pair_320x240_single_scale_on_the_oracle():
    f0, f1 = load_pair((320, 240))
    assert f0.shape == (3, 240, 320) and f0.max() <= 255 and f0.min() >= 0 and (f0 == np.round(f0)).all()
    res = rp.dense_flow_oracle(f0, f1, 17, 17, 7, 7)
    assert res["cost"].shape == (218, 298, 17, 17)                      # 75 MB
    assert res["idx"].min() >= 1 and res["idx"].max() <= 289
    assert np.abs(res["fy"]).max() <= 8 and np.abs(res["fx"]).max() <= 8
    via = rp.dense_flow_oracle(f0, f1, 17, 17, 7, 7, via_unfold=True)   # unfold -> crop -> SpatialMatching == the fused loop nest
    assert np.array_equal(via["cost"], res["cost"]) and np.array_equal(via["idx"], res["idx"])
    assert res["flowp"].shape == (4, 240, 320)
    assert (res["flowp"][:, :11] == 0).all() and (res["flowp"][:, :, :11] == 0).all()   # pad-back border floor(16/2)+floor(6/2)
    assert (res["flowp"][2] > 0).mean() > 0.3


def celiu_field(half=True):
    """celiu/output/car_flow.jpg decoded: (U, V, R) = direction * relative magnitude and the relative magnitude, at 640x480 or as 2 x 2
    block means at 320x240"""
    from PIL import Image

    from tests import flowcolor as fc

    img = np.asarray(Image.open(os.path.join(CELIU, "car_flow.jpg")).convert("RGB"))
    assert img.shape == (480, 640, 3)
    ux, uy, rad = fc.decode(img)
    U, V = ux * rad, uy * rad
    if half:
        U, V = U.reshape(240, 2, 320, 2).mean(axis=(1, 3)), V.reshape(240, 2, 320, 2).mean(axis=(1, 3))
    return U, V, np.hypot(U, V)


def agreement_with_celiu(fy, fx, conf, U, V, R):
    """fraction of the confident, moving pixels (celiu's relative magnitude > 0.1: below that the hue of a JPEG pixel says little) whose
    block-matching displacement lies within 45 degrees of celiu's; mean x-displacement inside / outside the fast blob (R > 0.6)"""
    mag = np.hypot(fx, fy)
    m = conf & (R > 0.1) & (mag > 0)
    cosang = (fx * U + fy * V) / np.maximum(mag * R, 1e-9)
    car = R > 0.6
    return float((cosang[m] > np.cos(np.pi / 4)).mean()), float(fx[car & conf].mean()), float(fx[~car & conf].mean()), int(m.sum()), int((car & conf).sum())


def test_flow_colour_decoder_round_trip():
    """tests/flowcolor.py against its own forward restatement of computeColor.m on a random field: direction within 5 degrees (the
    colour quantisation), relative magnitude within 0.005"""
    from tests import flowcolor as fc

    rng = np.random.default_rng(0)
    ang, r = rng.uniform(-np.pi, np.pi, (60, 80)), rng.uniform(0.1, 1.0, (60, 80))
    img = fc.encode(r * np.cos(ang), r * np.sin(ang))
    ux, uy, rad = fc.decode(img)
    err = np.degrees(np.abs(np.angle((ux + 1j * uy) / np.exp(1j * ang))))
    assert err.max() < 5 and np.abs(rad - r).max() < 0.005
    wheel = fc.make_colorwheel()
    assert wheel.shape == (55, 3) and (wheel.min(axis=1) == 0).all() and (wheel.max(axis=1) == 255).all()


def test_car_pair_agrees_with_the_celiu_flow_the_reference_ships():
    """north_star: "outputs match ... the celiu Coarse2FineTwoFrames ground truth".  Block matching and a variational flow are different
    estimators, so the comparison is the one the data allows: where celiu sees motion, the block-matching displacement of the confident
    pixels points the same way (within 45 degrees) for at least 70 % of them (measured: 93 %), and the blob celiu paints as the
    fastest -- the car, moving left -- is where block matching finds its large leftward displacements (mean x-flow below -3 px at
    half resolution, against a background that drifts slightly to the right)."""
    f0, f1 = load_pair((320, 240))
    res = rp.dense_flow_oracle(f0, f1, 17, 17, 7, 7)
    U, V, R = celiu_field(half=True)
    frac, fx_car, fx_bg, n, ncar = agreement_with_celiu(res["flowp"][0], res["flowp"][1], res["flowp"][2] > 0, U, V, R)
    assert n > 30000 and ncar > 3000
    assert frac >= 0.70, frac
    assert fx_car < -3.0 and -1.0 < fx_bg < 1.5, (fx_car, fx_bg)
    # celiu's own picture of the pair: the background drifts right and slightly down, the car moves left
    bg = (R > 0.1) & (R < 0.3)
    assert U[bg].mean() > 0 and U[R > 0.6].mean() < 0
