"""Regenerates tests/golden/golden_v1.npz.

Provenance: the reference (Lua/Torch7 + native files needing luaT.h/TH.h) cannot be imported,
compiled or run in the build image, so these vectors are NOT outputs of the reference.  They are
outputs of the CPU oracle (oracle/dfe_oracle.c) on seeded inputs, frozen so that (a) the oracle
cannot drift silently and (b) the GPU tests have inputs/expected outputs that travel without the
oracle being rebuilt identically.  What pins the oracle to the reference are the data-free
known-answer tests in tests/test_oracle_cpu.py.  Run: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import oracle as orc  # noqa: E402
from tests import refpath as rp  # noqa: E402

out = {}
im1, im2, fb, (hK, wK, hW, wW) = rp.kat_case(3, seed=5, C=3, h=24, w=28)
res = rp.dense_flow_oracle(im1, im2, hW, wW, hK, wK)
out.update(kat3_im1=im1, kat3_im2=im2, kat3_flowbase=fb, kat3_geo=np.array([hK, wK, hW, wW]))
for k in ("cost", "idx", "best", "scores", "imaxs"):
    out["kat3_" + k] = res[k]
ids = np.arange(1, 161, dtype=np.int64)
_, y, x = orc.x2yx_multi(8, 8, [1, 2, 4], ids)
out.update(codec_8_8_124_y=y, codec_8_8_124_x=x)
yc, xc = orc.x2yx_multi_compat_c(8, 8, [1, 2, 4], ids, fill=-999)
out.update(codec_compat_y=yc, codec_compat_x=xc)
# integer-valued frame pair (bit-exact regime) with 7x7 patch, 9x9 window
f0, f1, flow, foe = rp.synth_pair(48, 56, C=3, seed=11, max_flow=3)
r = rp.dense_flow_oracle(f0, f1, 9, 9, 7, 7)
out.update(int_f0=f0, int_f1=f1, int_cost=r["cost"], int_idx=r["idx"], int_scores=r["scores"], int_imaxs=r["imaxs"])
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v1.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
