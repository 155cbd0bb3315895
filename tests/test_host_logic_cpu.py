"""Host logic that needs no GPU: the version2 script's geometry table and decode (version2/test.lua:6-21, 45-51), the flat parameter
vector of getNetwork (test.lua:40-42), prepareInput's narrow (opticalflow_model.lua:131-151) and the option-key table of the C ABI."""
import math
import re
import os

import numpy as np
import pytest
import torch

import depth_estimation_amd as dfe

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_version2_datap_follows_the_script():
    """datap.hKernel / wKernel = 1 + sum(k - 1); lWin = ceil(wWin / 2) - 1, tWin = ceil(hWin / 2) - 1, rWin = floor(wWin / 2),
    bWin = floor(hWin / 2) (version2/test.lua:14-21), for odd and even windows."""
    d = dfe.version2.defaultDatap()
    assert (d["wImg"], d["hImg"], d["normalization_k"], d["wWin"], d["hWin"]) == (320, 180, 17, 17, 17)
    assert d["hKernel"] == 17 and d["wKernel"] == 17 and (d["lWin"], d["tWin"], d["rWin"], d["bWin"]) == (8, 8, 8, 8)
    e = dfe.version2.defaultDatap(layers=((3, 5, 7, 4), (4, 3, 5, 8)), wWin=16, hWin=10)
    assert e["hKernel"] == 1 + 6 + 4 and e["wKernel"] == 1 + 4 + 2
    assert (e["lWin"], e["rWin"], e["tWin"], e["bWin"]) == (7, 8, 4, 5)
    assert e["lWin"] + e["rWin"] == 16 - 1 and e["tWin"] + e["bWin"] == 10 - 1


def test_version2_decode_is_the_first_minimum_of_the_window():
    """decodeFlow (test.lua:45-51): idx - 1 over the flat window, yflow = floor(idx / wWin) - tWin, xflow = idx % wWin - lWin; ties go
    to the first cell in index order."""
    rng = np.random.default_rng(0)
    H, W, hWin, wWin = 6, 9, 5, 7
    vol = rng.integers(0, 4, size=(H, W, hWin, wWin)).astype(np.float32)        # many ties
    d = dfe.version2.defaultDatap(wWin=wWin, hWin=hWin)
    xf, yf = dfe.version2.decodeFlow(torch.from_numpy(vol), d)
    flat = vol.reshape(H, W, -1)
    idx = flat.argmin(axis=2)                                                   # numpy: first minimum
    assert np.array_equal(yf.numpy(), idx // wWin - d["tWin"]) and np.array_equal(xf.numpy(), idx % wWin - d["lWin"])
    assert xf.dtype == torch.int64 and yf.dtype == torch.int64


def test_version2_reshape_module_round_trip():
    m = dfe.version2.Reshape(35)
    x = torch.arange(35, dtype=torch.float32).reshape(1, 1, 5, 7)
    y = m.forward(x)
    assert tuple(y.shape) == (35,) and torch.equal(y, x.reshape(35))
    g = m.updateGradInput(x, torch.ones(35))
    assert tuple(g.shape) == (1, 1, 5, 7)


def test_prepare_input_shapes_without_a_device():
    """prepareInput's narrow of patch 1 (opticalflow_model.lua:143-149) is pure tensor slicing: rows / columns ceil(max / 2) (1-based) ..,
    H - maxh + 1 of them; the multiscale graph takes both patches as they are."""
    a, b = torch.arange(3 * 40 * 52, dtype=torch.float32).reshape(3, 40, 52), torch.ones((3, 40, 52))
    geo = dict(layers=[[3, 5, 5, 4]], maxh=16, maxw=17, multiscale=False)
    p1, p2 = dfe.prepareInput(geo, a, b)
    assert p2 is b and tuple(p1.shape) == (3, 40 - 16 + 1, 52 - 17 + 1)
    assert torch.equal(p1, a[:, math.ceil(16 / 2) - 1 : math.ceil(16 / 2) - 1 + 25, math.ceil(17 / 2) - 1 : math.ceil(17 / 2) - 1 + 36])
    m1, m2 = dfe.prepareInput(dict(layers=[[3, 5, 5, 4]], maxh=8, maxw=8, multiscale=True), a, b)
    assert m1 is a and m2 is b


def test_option_keys_are_the_header_s_and_the_table_s():
    """The option keys of dfe_set_option: the list in include/dfe.h, the table in csrc/dfe_ctx.hip and the host mirror's OPTION_KEYS agree."""
    from depth_estimation_amd._lib import OPTION_KEYS

    hdr = open(os.path.join(ROOT, "include", "dfe.h")).read()
    doc = hdr[hdr.index("Behaviour switches of the launchers") : hdr.index("int dfe_set_option")]
    in_header = set(re.findall(r'"([a-z0-9_]+)"', doc))
    src = open(os.path.join(ROOT, "depth-estimation_amd", "csrc", "dfe_ctx.hip")).read()
    table = src[src.index("const DfeOptName dfe_opt_names[DFE_NOPT] = {") : src.index("};", src.index("const DfeOptName dfe_opt_names[DFE_NOPT] = {"))]
    in_table = set(re.findall(r'\{"([a-z0-9_]+)",', table)) | {"graphs"}
    assert set(OPTION_KEYS) == in_header == in_table, (sorted(set(OPTION_KEYS) ^ in_header), sorted(in_header ^ in_table))
    enum = open(os.path.join(ROOT, "depth-estimation_amd", "csrc", "dfe_internal.h")).read()
    n_enum = len(re.findall(r"^\s+DFE_OPT_[A-Z0-9_]+", enum[enum.index("enum DfeOpt {") : enum.index("DFE_NOPT")], flags=re.M))
    assert n_enum == len(in_table) - 1
