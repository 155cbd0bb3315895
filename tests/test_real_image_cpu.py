"""CPU suite, BASELINE configs[0]: the celiu car1 -> car2 pair (tests/golden/celiu/*.jpg are byte copies of
/root/reference/celiu/car1.jpg, car2.jpg -- input data), halved to 320x240 as celiu/demoflow.m:11-12 does, through the
single-scale dense path on the CPU oracle: 7x7 patch, +-8 search (17x17), i.e. the tests/time_matching.lua /
compute_cartesian_groundtruth_cross_correlation path (unfold -> SpatialMatching -> min + centre tie-break -> decode ->
extractOutput -> pad back).  Plumbing: shapes, ranges, the unfold route == the fused route, and agreement with the
variational flow the reference ships as its accuracy comparator (Coarse2FineFlow on this pair has mean
(vx, vy) = (-0.74, 1.67) px at 640x480, max 15.3 px, i.e. about (-0.4, 0.8) at half size)."""
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
    # the scene moves the way the reference's variational comparator says: mostly downwards by ~1 px, slightly left
    conf = res["flowp"][2] > 0
    fy, fx = res["flowp"][0][conf], res["flowp"][1][conf]
    assert conf.mean() > 0.3
    assert 0.2 < fy.mean() < 2.0 and -1.5 < fx.mean() < 0.5
    assert (np.abs(fy - 0.8) <= 2).mean() > 0.7                         # most confident pixels within 2 px of it
