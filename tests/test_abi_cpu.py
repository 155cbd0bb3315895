"""CPU suite: the C-ABI library loads and exports every symbol include/dfe.h declares; the
product refuses to run without a device; the product never touches the oracle."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "dfe.h")
PKG = os.path.join(ROOT, "depth-estimation_amd")


def declared_symbols():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dfe_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound(dfe):
    from depth_estimation_amd import _lib

    syms = declared_symbols()
    assert len(syms) >= 20
    l = dfe.lib()
    for s in syms:
        assert hasattr(l, s), "libdfe.so does not export %s" % s
        assert s in _lib.PROTOTYPES, "no ctypes prototype for %s" % s
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (dfe_[a-z0-9_]+)", nm))
    assert set(syms) <= exported
    assert exported - set(syms) == set(), "exported but undeclared: %s" % (exported - set(syms))


def test_lua_ffi_binding_declares_the_whole_abi():
    """The LuaJIT-FFI stub a maintainer adds on the reference side declares every entry point of
    include/dfe.h, and its cdef block is the one tools/gen_lua_cdef.py derives from the header (no drift)."""
    lua = open(os.path.join(PKG, "lua", "dfe_ffi.lua")).read()
    cdef = lua[lua.index("ffi.cdef[[") : lua.index("]]")]
    for s in declared_symbols():
        assert re.search(r"\b%s\s*\(" % s, cdef), "dfe_ffi.lua does not declare %s" % s
    rc = subprocess.run(["python", os.path.join(ROOT, "tools", "gen_lua_cdef.py"), "--check"]).returncode
    assert rc == 0, "dfe_ffi.lua is out of date: run tools/gen_lua_cdef.py"


def test_library_is_gfx950_only():
    from depth_estimation_amd import _lib

    blob = open(_lib.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_no_device_fails_loudly(dfe):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(dfe.DfeError) as e:
        dfe.Context(0)
    assert "no CPU fallback" in str(e.value)
    with pytest.raises(dfe.DfeError):
        dfe.nn.SpatialMatching(3, 3).forward([torch.zeros(2, 4, 4), torch.zeros(2, 6, 6)])


def test_host_scalar_codec_matches_oracle(dfe, oracle):
    # dfe_yx2x_multi / dfe_x2yx_multi_number are host-side (no device work): check against the oracle
    for maxh, maxw, ratios in [(8, 8, [1, 2, 4]), (4, 4, [1, 2, 4, 8]), (16, 16, [1, 2, 4, 8]), (8, 8, [1, 2])]:
        geo = dict(maxh=maxh, maxw=maxw, ratios=ratios, multiscale=True)
        n = oracle.multi_nclasses(maxh, maxw, ratios)
        r, nr = dfe._lib.ratios_array(ratios)
        assert dfe.lib().dfe_multi_nclasses(maxh, maxw, r, nr) == n
        for i in range(1, n + 1):
            rc, y, x = oracle.x2yx_multi_number(maxh, maxw, ratios, i)
            assert dfe.x2yxMultiNumber(geo, i) == (y, x)
            assert dfe.yx2xMulti(geo, y, x) == i == oracle.yx2x_multi(maxh, maxw, ratios, y, x)
        with pytest.raises(AssertionError):
            dfe.x2yxMultiNumber(geo, n + 1)
        assert dfe.getMiddleIndex(geo) == oracle.yx2x_multi(maxh, maxw, ratios, 0, 0)
    assert dfe.getMiddleIndex(dict(maxh=17, maxw=17)) == 9 + 17 * 8
    assert dfe.getMiddleIndex(dict(maxh=16, maxw=16)) == 8 + 16 * 7


def test_product_does_not_reference_oracle():
    bad = []
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".lua", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"dfe_oracle|libdfe_oracle|from tests|import tests|oracle/", txt):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, "product files reference the oracle: %s" % bad


def test_committed_traffic_summary_belongs_to_the_committed_kernel():
    """profiles/traffic_vga.json (PMC-derived HBM bytes that bench.py reports as roofline.traffic) is tagged with the kernel
    revision it was measured on; bench.py drops it when the library reports another one.  A kernel change that bumps
    DFE_CV_KERNEL_REV without new PMC passes shows up here instead of as a silent `"traffic": null` in the bench line."""
    import json
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "depth-estimation_amd", "csrc", "ssd_cost_volume.hip")).read()
    rev = re.search(r'#define DFE_CV_KERNEL_REV "([^"]+)"', src).group(1)
    traffic = json.load(open(os.path.join(root, "profiles", "traffic_vga.json")))
    assert traffic["kernel_rev"] == rev, "profiles/traffic_vga.json is from %s, the source is %s: rerun tools/pmc_cv.sh + tools/make_traffic.py" % (traffic["kernel_rev"], rev)
    assert traffic["hbm_bytes_per_launch"] >= traffic["algorithmic_bytes_per_launch"]
