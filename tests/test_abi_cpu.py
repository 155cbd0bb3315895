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


def test_lua_cdef_block_is_valid_c(tmp_path):
    """LuaJIT parses the ffi.cdef block as C declarations and rejects ALL of it on the first unknown type (round 2 shipped a block
    that used dfe_radial_params without declaring it: every Lua drop-in failed at require).  A C compiler must accept the
    block as it stands, with only the two headers whose types LuaJIT predeclares (stddef / stdint), and every struct of
    include/dfe.h must be declared in it with the header's layout."""
    lua = open(os.path.join(PKG, "lua", "dfe_ffi.lua")).read()
    cdef = lua[lua.index("ffi.cdef[[") + len("ffi.cdef[[") : lua.index("]]")]
    src = tmp_path / "cdef.c"
    src.write_text("#include <stddef.h>\n#include <stdint.h>\n" + cdef + "\n")
    r = subprocess.run(["gcc", "-fsyntax-only", "-std=c99", "-Wall", "-Werror", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    hdr = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    structs = re.findall(r"typedef\s+struct\s+(\w+)\s*\{", hdr)
    assert "dfe_radial_params" in structs and "dfe_filter_layer" in structs
    for name in structs:
        assert re.search(r"typedef struct %s \{" % name, cdef), "dfe_ffi.lua does not declare struct %s" % name
    # same layout as the header: sizeof / offsetof through a C program that includes both (the cdef's copy under other names)
    prog = tmp_path / "layout.c"
    renamed = re.sub(r"\bdfe_", "lua_dfe_", cdef)
    checks = "".join("_Static_assert(sizeof(%s) == sizeof(lua_%s), \"%s\");\n" % (n, n, n) for n in structs)
    prog.write_text('#include "%s"\n%s\n%s' % (HDR, renamed, checks))
    r = subprocess.run(["gcc", "-fsyntax-only", "-std=c11", str(prog)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_shared_filter_clone_follows_rebound_parameters():
    """filter:clone('weight','bias','gradWeight','gradBias'): the second branch reads the first one's parameters at call time --
    rebinding them on the source (how trained weights are loaded), switching its kernel, or assigning through the clone all
    act on ONE set of tensors (the clone used to snapshot the tensors at construction)."""
    import torch

    from depth_estimation_amd import network

    gen = torch.Generator().manual_seed(1)
    filt = network.getFilter(dict(layers=[(3, 5, 5, 4), (4, 3, 3, 6)]), device="cpu", generator=gen)
    shared = network._SharedFilter(filt)
    src, cl = filt.modules[0], shared.modules[0]
    assert type(cl).__name__ == "SharedSpatialConvolution" and isinstance(cl, network.SpatialConvolution)
    assert cl.weight is src.weight and cl.gradBias is src.gradBias and cl.nOutputPlane == 4
    w2 = torch.randn_like(src.weight)
    src.weight = w2                                    # rebinding on branch 1 ...
    assert cl.weight is w2                             # ... is what branch 2 computes with
    src.kernel = "mfma"
    assert cl.kernel == "mfma"
    cl.bias = torch.zeros(4)                           # and assignment through the clone lands in the source
    assert src.bias is cl.bias and float(src.bias.abs().sum()) == 0
    cl.output = torch.ones(1)                          # outputs stay the clone's own
    assert src.output is None
    assert set(shared.getWeights()) == {"layer1", "layer2"} and shared.getWeights()["layer1"] is w2


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
    import sys

    sys.path.insert(0, os.path.join(root, "tools"))
    from kernel_hash import kernel_source_hash

    assert traffic.get("source_sha256") == kernel_source_hash(), (
        "profiles/traffic_vga.json was measured on another version of the cost-volume kernels: rerun tools/refresh_traffic.sh on the GPU "
        "(and bump DFE_CV_KERNEL_REV if the change alters what the kernels read or write)")
    assert traffic["kernel_rev"] == rev, "profiles/traffic_vga.json is from %s, the source is %s: rerun tools/pmc_cv.sh + tools/make_traffic.py" % (traffic["kernel_rev"], rev)
    assert traffic["hbm_bytes_per_launch"] >= traffic["algorithmic_bytes_per_launch"]
