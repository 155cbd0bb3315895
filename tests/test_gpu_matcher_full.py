"""GPU suite (-m gpu), part 3: the flat-tile feature matcher (csrc/feat_matching_flat.hip) AT THE SIZES THE BENCHMARKS RUN.

The kernel is persistent: one block per CU walks tiles r * gridDim + vb, the virtual block index is permuted per XCD when the grid is
a multiple of 8, the three staging buffers rotate across tiles and the copy-out image of tile t is rebuilt while the stores of tile
t - 1 drain.  Shapes of <= 256 tiles (everything in test_gpu_parity.py) never run a second loop iteration; the shapes below run 4 - 14
rounds with the permutation active.  Each case: output pre-filled with NaN, the kernel named, compared BITWISE with the round-3 row
kernels (fm_flat = 0) over the whole output on the device, and with the ORACLE on three row bands (first rows, a band straddling the
first round boundary, last rows) on the host.  (nn.SpatialMatching: version2/network.lua:30, opticalflow_model.lua:93,
tests/time_matching.lua:18.)"""
import numpy as np
import pytest
import torch

from tests import oracle as orc
from tests import refpath as rp

pytestmark = pytest.mark.gpu


def T(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def _bands(H1, W1, ncu, rows=3):
    """first rows, the rows around the first round boundary (tile ncu starts at pixel group 64 * ncu), last rows"""
    G = -(-W1 // 4)
    yb = min(max((64 * ncu) // G, rows), H1 - rows)
    return sorted({(0, rows), (yb - rows // 2 - 1, yb + rows // 2 + 1), (H1 - rows, H1)})


def _first_min_index(vol):
    """[H][W][h][w] device volume -> int64 0-based index of the FIRST minimum of every window (exact, whatever torch.argmin does on ties)"""
    H1, W1 = vol.shape[:2]
    v = vol.reshape(H1, W1, -1)
    out = torch.empty((H1, W1), dtype=torch.int64, device=vol.device)
    ar = torch.arange(v.shape[2], device=vol.device)
    for y0 in range(0, H1, 64):                                       # (row blocks: the comparison's temporaries stay small)
        blk = v[y0 : y0 + 64]
        eq = blk == blk.min(dim=2, keepdim=True).values
        out[y0 : y0 + 64] = torch.where(eq, ar, v.shape[2]).min(dim=2).values
    return out


BENCHED = [
    (32, 448, 608, 17, 17),     # version2's matcher at VGA (bench workload version2-vga): 1065 tiles, 4.16 rounds, the extra-row task
    (32, 465, 625, 16, 16),     # nn.SpatialMatching(16, 16) on 32 planes at VGA: 1141 tiles, W1 % 4 = 1 (ragged last group of every row)
    (10, 465, 625, 16, 16),     # ... on time_matching.lua's 10 planes (quarter tiles by default)
    (32, 705, 1265, 16, 16),    # 720p: 3490 tiles, 13.6 rounds
]


@pytest.mark.parametrize("K,H1,W1,mh,mw", BENCHED)
def test_flat_matcher_at_benched_sizes_bitwise_and_oracle_bands(dfe, cuda, K, H1, W1, mh, mw):
    ctx = dfe.get_ctx(0)
    lib = dfe.lib()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    ntiles = -(-(H1 * -(-W1 // 4)) // 64)
    assert ntiles >= 4 * ncu and ncu % 8 == 0, "the shape must run several rounds per block with the XCD permutation active"
    rng = np.random.default_rng(K * 7 + W1 + mh)
    in1 = rng.standard_normal((K, H1, W1)).astype(np.float32)
    in2 = rng.standard_normal((K, H1 + mh - 1, W1 + mw - 1)).astype(np.float32)
    t1, t2 = T(in1, cuda), T(in2, cuda)
    with ctx.options(fm_flat=0):
        old = torch.full((H1, W1, mh, mw), float("nan"), device=cuda)
        ctx.check(lib.dfe_spatial_matching_f32(ctx.handle, t1.data_ptr(), t2.data_ptr(), K, H1, W1, mh, mw, old.data_ptr()))
        assert ctx.last_kernel() != "feat_matching_flat_kernel"
    assert not bool(torch.isnan(old).any())
    splits = (-1,) if mh == 17 else (-1, 0, 2, 4)                     # the launcher's own choice; whole tiles / halves / quarters
    for split in splits:
        with ctx.options(fm_split=split):
            out = torch.full((H1, W1, mh, mw), float("nan"), device=cuda)
            ctx.check(lib.dfe_spatial_matching_f32(ctx.handle, t1.data_ptr(), t2.data_ptr(), K, H1, W1, mh, mw, out.data_ptr()))
            torch.cuda.synchronize()
            assert ctx.last_kernel() == "feat_matching_flat_kernel", ctx.last_kernel()
        nan = int(torch.isnan(out).sum())
        assert nan == 0, "fm_split %d: %d cells left unwritten" % (split, nan)
        if not torch.equal(out, old):
            bad = (out != old).reshape(H1, W1, -1).any(dim=2).nonzero()
            pytest.fail("fm_split %d: %d pixels differ from the row kernel, first at (y, x) = %s" % (split, bad.shape[0], bad[0].tolist()))
        if split == -1:
            for y0, y1 in _bands(H1, W1, ncu):
                ref = orc.spatial_matching(np.ascontiguousarray(in1[:, y0:y1]), np.ascontiguousarray(in2[:, y0 : y1 + mh - 1]), mh, mw)
                assert np.array_equal(out[y0:y1].cpu().numpy(), ref), "rows %d..%d differ from the oracle" % (y0, y1)
        del out


@pytest.mark.parametrize("K,H1,W1,mh,mw", BENCHED[:2] + [(8, 300, 420, 7, 16), (8, 300, 420, 4, 17)])
def test_flat_matcher_argmin_form_at_benched_sizes(dfe, cuda, K, H1, W1, mh, mw):
    """dfe_spatial_matching_argmin_f32 (matcher + first-minimum decode in one kernel, no volume; version2/test.lua:45-51): the index of
    every pixel equals the first minimum of the volume form's window (whole output, on the device) and the oracle's on three row bands;
    the decoded flows are the index's.  The two short windows (4 and 7 rows: fewer than 8) are the shapes whose candidate arrays are
    larger than the copy-out image they share LDS with (round-4 advisor finding)."""
    ctx = dfe.get_ctx(0)
    lib = dfe.lib()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    rng = np.random.default_rng(K + H1 + mh)
    in1 = rng.standard_normal((K, H1, W1)).astype(np.float32)
    in2 = rng.standard_normal((K, H1 + mh - 1, W1 + mw - 1)).astype(np.float32)
    t1, t2 = T(in1, cuda), T(in2, cuda)
    idx = torch.full((H1, W1), -7, dtype=torch.int64, device=cuda)
    xf, yf = torch.full((H1, W1), float("nan"), device=cuda), torch.full((H1, W1), float("nan"), device=cuda)
    ctx.check(lib.dfe_spatial_matching_argmin_f32(ctx.handle, t1.data_ptr(), t2.data_ptr(), K, H1, W1, mh, mw, idx.data_ptr(), xf.data_ptr(), yf.data_ptr()))
    torch.cuda.synchronize()
    assert ctx.last_kernel() == "feat_matching_flat_kernel+argmin", ctx.last_kernel()
    vol = torch.empty((H1, W1, mh, mw), device=cuda)
    ctx.check(lib.dfe_spatial_matching_f32(ctx.handle, t1.data_ptr(), t2.data_ptr(), K, H1, W1, mh, mw, vol.data_ptr()))
    want = _first_min_index(vol)
    if not torch.equal(idx, want + 1):
        bad = (idx != want + 1).nonzero()
        pytest.fail("%d pixels differ from the volume's first minimum, first at (y, x) = %s: %d vs %d" % (
            bad.shape[0], bad[0].tolist(), int(idx[tuple(bad[0])]), int(want[tuple(bad[0])]) + 1))
    lWin, tWin = (mw + 1) // 2 - 1, (mh + 1) // 2 - 1
    assert torch.equal(yf, (want // mw - tWin).to(torch.float32)) and torch.equal(xf, (want % mw - lWin).to(torch.float32))
    for y0, y1 in _bands(H1, W1, ncu):
        ref = orc.spatial_matching(np.ascontiguousarray(in1[:, y0:y1]), np.ascontiguousarray(in2[:, y0 : y1 + mh - 1]), mh, mw)
        assert np.array_equal(idx[y0:y1].cpu().numpy(), ref.reshape(y1 - y0, W1, -1).argmin(axis=2) + 1), "rows %d..%d" % (y0, y1)
    # the fallback (the round-3 matcher into the scratch arena + the decode kernel) gives the same
    with ctx.options(fm_flat=0):
        idx2 = torch.empty_like(idx)
        ctx.check(lib.dfe_spatial_matching_argmin_f32(ctx.handle, t1.data_ptr(), t2.data_ptr(), K, H1, W1, mh, mw, idx2.data_ptr(), None, None))
        assert ctx.last_kernel() != "feat_matching_flat_kernel+argmin"
    assert torch.equal(idx2, idx)


@pytest.mark.parametrize("hWin,wWin", [(4, 16), (7, 17), (4, 17), (7, 16)])
def test_version2_short_windows_lean_equals_volume_path(dfe, cuda, hWin, wWin):
    """dfe_version2_flow_pair_f32 with windows of 4 .. 7 rows (round-4 advisor: the arg-min form's candidates overran the LDS image sized
    for the copy-out): volume = NULL (matcher + arg-min kernel) == volume != NULL (volume kernel + decode), index and flows."""
    v2 = dfe.version2
    H, W = 40, 300 + wWin + 4
    datap = v2.defaultDatap(wImg=W, hImg=H, normalization_k=5, layers=[(3, 5, 5, 6)], wWin=wWin, hWin=hWin)
    g = torch.Generator().manual_seed(hWin * 20 + wWin)
    net = v2.getNetwork(datap, device=cuda, generator=g)
    rng = np.random.default_rng(hWin + wWin)
    prev = rng.random((3, H, W), dtype=np.float32)
    cur = np.roll(prev, (1, -2), axis=(1, 2)) + rng.normal(0, 0.01, (3, H, W)).astype(np.float32)
    tp, tc = T(prev, cuda), T(cur, cuda)
    full = v2.flowPair(net, datap, tp, tc, one_call=True, want_volume=True)
    lean = v2.flowPair(net, datap, tp, tc, one_call=True, want_volume=False)
    assert dfe.get_ctx(0).last_kernel() == "feat_matching_flat_kernel+argmin"
    for k in ("index", "xflow", "yflow"):
        assert torch.equal(lean[k], full[k]), k
    assert torch.equal(full["index"], _first_min_index(full["volume"]) + 1)


def test_version2_one_call_at_vga_32_planes_against_the_oracle_on_bands(dfe, cuda):
    """The bench workload `version2-vga` as it runs: 480 x 640 frames, normalisation 17, one 17 x 17 layer of 32 planes, 17 x 17 window --
    conv_batch_kernel<17, 8> on 32 planes and 4.2 rounds of the matcher.  One call (lean and with the volume) == the staged module path
    bit for bit; against the oracle composition on three bands of output rows (normalisation of the whole frames, convolution and
    matcher on the rows a band needs: both are 'valid' operators, so a band's values are the full frame's): volume 1e-4 relative (the
    normalisation divides by a device sqrt), indices wherever the oracle's two best costs are further apart than that."""
    v2 = dfe.version2
    H, W = 480, 640
    datap = v2.defaultDatap(wImg=W, hImg=H)
    assert [tuple(l) for l in datap["layers"]] == [(3, 17, 17, 32)] and datap["hWin"] == 17 and datap["normalization_k"] == 17
    g = torch.Generator().manual_seed(3)
    net = v2.getNetwork(datap, device=cuda, generator=g)
    flat = (torch.rand(net.flatParameters().numel(), generator=g) - 0.5) * 0.1
    net.loadParameters(flat)
    f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=5, max_flow=7, noise_sigma=1.0)
    prev, cur = f0 / np.float32(255), f1 / np.float32(255)
    tp, tc = T(prev, cuda), T(cur, cuda)
    staged = v2.flowPair(net, datap, tp, tc, one_call=False, want_volume=True)
    one = v2.flowPair(net, datap, tp, tc, one_call=True, want_volume=True)
    lean = v2.flowPair(net, datap, tp, tc, one_call=True, want_volume=False)
    assert dfe.get_ctx(0).last_kernel() == "feat_matching_flat_kernel+argmin"
    for k in ("volume", "index", "xflow", "yflow"):
        assert torch.equal(staged[k], one[k]), k
    for k in ("index", "xflow", "yflow"):
        assert torch.equal(lean[k], one[k]), k
    H1, W1 = one["index"].shape
    assert (H1, W1) == (448, 608)
    conv = [m for m in net.modules[0].modules[0].modules[2:]]
    ws, bs = [m.weight.cpu().numpy() for m in conv], [m.bias.cpu().numpy() for m in conv]
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    nclear = 0
    for y0, y1 in _bands(H1, W1, ncu, rows=2):
        ref = rp.version2_band_oracle(prev, cur, datap, ws, bs, y0, y1)
        vol = one["volume"][y0:y1].cpu().numpy()
        amax = float(np.abs(ref["volume"]).max())
        assert np.allclose(vol, ref["volume"], rtol=1e-4, atol=1e-5 * amax), "rows %d..%d" % (y0, y1)
        srt = np.sort(ref["volume"].reshape(y1 - y0, W1, -1), axis=2)
        clear = (srt[..., 1] - srt[..., 0]) > 2e-4 * srt[..., 1] + 1e-5 * amax
        nclear += int(clear.sum())
        assert clear.mean() > 0.8
        for k in ("index", "xflow", "yflow"):
            assert np.array_equal(lean[k][y0:y1].cpu().numpy()[clear], ref[k][clear]), (k, y0)
    assert nclear > 2000


def test_learned_pyramid_above_1p5_megapixels_reference_order_equals_staged(dfe, cuda):
    """Round-3 advisor: with dfe_set_cost_volume_kernel(1) (reference summation order) the multiscale launcher planned a fused second
    scale that the feature matcher then refused; the plan now uses the launcher's own predicate (dfe_feat_matching_win64_ok).  At the
    size that triggered it (>= 1.5 MP, where the fused scales are planned): one call == staged bit for bit with the reference-order
    kernels forced, and == the default kernels' result (the feature matchers sum in the same order either way)."""
    H, W, ratios = 1024, 1536, [1, 2, 4, 8]
    layers = [(3, 5, 5, 4), (4, 5, 5, 4), (4, 5, 5, 10)]
    gen = torch.Generator().manual_seed(4)
    geo = dict(maxh=8, maxw=8, ratios=ratios, multiscale=True, layers=layers, share_filters=True, hImg=H, wImg=W, output_extraction_method="max")
    model = dfe.getModelMultiscale(geo, True, False, device=cuda, generator=gen)
    f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=9, max_flow=12, noise_sigma=0)
    t0, t1 = T(f0 / np.float32(255), cuda), T(f1 / np.float32(255), cuda)
    ctx = dfe.get_ctx(0)
    dflt = model.forwardFlow([t0, t1], False, one_call=True)
    ctx.set_cost_volume_kernel(1)
    try:
        one = model.forwardFlow([t0, t1], False, one_call=True)
        stg = model.forwardFlow([t0, t1], False, one_call=False)
    finally:
        ctx.set_cost_volume_kernel(0)
    for k in ("index", "y", "x"):
        assert torch.equal(one[k], stg[k]), k
        assert torch.equal(one[k], dflt[k]), k
    inner = (slice(150, -150), slice(150, -150))
    assert float((one["y"][inner] == dflt["y"][inner]).float().mean()) == 1.0


def test_version2_with_matrix_core_convolution_at_vga(dfe, cuda):
    """dfe_set_option("conv_mfma", 1): version2's 17 x 17 x 3 -> 32 layer of BOTH frames as one launch of the resident-weights implicit
    GEMM (v_mfma_f32_16x16x4_f32, csrc/conv_mfma.hip) inside dfe_version2_flow_pair_f32 at 480 x 640.  Its features are the fmaf
    chain in the reference's (input plane, ky, kx) order: the volume stays within 1e-4 relative of the exact path's (1e-5 on the
    features, squared differences on top), the flows agree wherever the exact volume's two best costs are not within that band."""
    v2 = dfe.version2
    H, W = 480, 640
    datap = v2.defaultDatap(wImg=W, hImg=H)
    g = torch.Generator().manual_seed(3)
    net = v2.getNetwork(datap, device=cuda, generator=g)
    f0, f1, _, _ = rp.synth_pair(H, W, C=3, seed=5, max_flow=7, noise_sigma=1.0)
    tp, tc = T(f0 / np.float32(255), cuda), T(f1 / np.float32(255), cuda)
    ctx = dfe.get_ctx(0)
    exact = v2.flowPair(net, datap, tp, tc, one_call=True, want_volume=True)
    with ctx.options(conv_mfma=1):
        mm = v2.flowPair(net, datap, tp, tc, one_call=True, want_volume=True)
        lean = v2.flowPair(net, datap, tp, tc, one_call=True, want_volume=False)
    for k in ("index", "xflow", "yflow"):
        assert torch.equal(lean[k], mm[k]), k
    ve, vm = exact["volume"], mm["volume"]
    amax = float(ve.abs().max())
    assert bool(((vm - ve).abs() <= 1e-4 * ve.abs() + 1e-5 * amax).all())
    srt = torch.sort(ve.reshape(ve.shape[0], ve.shape[1], -1), dim=2).values
    clear = (srt[..., 1] - srt[..., 0]) > 2e-4 * srt[..., 1] + 2e-5 * amax
    assert float(clear.float().mean()) > 0.8
    assert torch.equal(mm["index"][clear], exact["index"][clear])
    differ = int((mm["index"] != exact["index"]).sum())
    assert differ <= 0.02 * mm["index"].numel(), differ
    # both matrix-core forms: the convolution's epilogue leaves |a|^2, |b|^2 for the banded-GEMM matcher (two launches for the pair).  Against
    # the volume of the same features: the index is its first minimum wherever the two best costs are further apart than the matcher's band
    with ctx.options(conv_mfma=1, fm_mfma=1):
        both = v2.flowPair(net, datap, tp, tc, one_call=True, want_volume=False)
        assert ctx.last_kernel() == "fmm_kernel+argmin", ctx.last_kernel()
    srt = torch.sort(vm.reshape(vm.shape[0], vm.shape[1], -1), dim=2).values
    near = (srt[..., 1] - srt[..., 0]) <= 2 * (1e-5 * srt[..., 1].abs() + 1e-6 * float(vm.abs().max()))
    d2 = both["index"] != mm["index"]
    assert not bool((d2 & ~near).any()), int((d2 & ~near).sum())
    assert int(d2.sum()) <= 0.01 * d2.numel(), int(d2.sum())
    for k in ("xflow", "yflow"):
        assert torch.equal(both[k][~d2], mm[k][~d2]), k


@pytest.mark.parametrize("K,H1,W1,mh", [(32, 448, 608, 17), (32, 60, 301, 17), (10, 50, 270, 16), (8, 23, 40, 17), (5, 9, 17, 16)])
def test_matrix_core_matcher_within_tolerance_of_the_exact_kernels(dfe, cuda, K, H1, W1, mh):
    """dfe_set_option("fm_mfma", 1): nn.SpatialMatching(17, 17) / (16, 16) as a banded GEMM on the matrix cores (csrc/feat_matching_mfma.hip,
    |a|^2 + |b|^2 - 2 a.b with v_mfma_f32_16x16x4_f32).  Against the exact kernels: every cost within 1e-5 |c| + 1e-6 max|c| (stated in
    include/dfe.h; the oracle on row bands confirms the exact side), every cell written (NaN pre-fill), the arg-min form's index equal to the
    exact first minimum except where the exact volume's two best costs lie within twice that band -- the count of differing pixels is
    reported and bounded -- and the decoded flows are the index's.  Shapes: version2's VGA matcher (2128 tiles = 8.3 rounds per block),
    ragged widths / heights (W1 % 16, H1 % 8 != 0), K % 8 != 0 (zero-filled planes)."""
    ctx = dfe.get_ctx(0)
    lib = dfe.lib()
    mw = mh
    rng = np.random.default_rng(K + H1 + W1)
    in1 = rng.standard_normal((K, H1, W1)).astype(np.float32)
    in2 = rng.standard_normal((K, H1 + mh - 1, W1 + mw - 1)).astype(np.float32)
    # plant a shifted copy so that many windows hold a near-zero cost (the cancellation case) next to large ones
    in2[:, 3 : 3 + H1, 5 : 5 + W1] = in1 + 0.05 * rng.standard_normal((K, H1, W1)).astype(np.float32)
    t1, t2 = T(in1, cuda), T(in2, cuda)
    exact = torch.empty((H1, W1, mh, mw), device=cuda)
    ctx.check(lib.dfe_spatial_matching_f32(ctx.handle, t1.data_ptr(), t2.data_ptr(), K, H1, W1, mh, mw, exact.data_ptr()))
    assert not ctx.last_kernel().startswith("fmm_kernel")
    for y0, y1 in [(0, min(2, H1)), (max(H1 - 2, 0), H1)]:
        ref = orc.spatial_matching(np.ascontiguousarray(in1[:, y0:y1]), np.ascontiguousarray(in2[:, y0 : y1 + mh - 1]), mh, mw)
        assert np.array_equal(exact[y0:y1].cpu().numpy(), ref)
    with ctx.options(fm_mfma=1):
        vol = torch.full((H1, W1, mh, mw), float("nan"), device=cuda)
        ctx.check(lib.dfe_spatial_matching_f32(ctx.handle, t1.data_ptr(), t2.data_ptr(), K, H1, W1, mh, mw, vol.data_ptr()))
        assert ctx.last_kernel() == "fmm_kernel", ctx.last_kernel()
        idx = torch.full((H1, W1), -7, dtype=torch.int64, device=cuda)
        xf, yf = torch.full((H1, W1), float("nan"), device=cuda), torch.full((H1, W1), float("nan"), device=cuda)
        ctx.check(lib.dfe_spatial_matching_argmin_f32(ctx.handle, t1.data_ptr(), t2.data_ptr(), K, H1, W1, mh, mw, idx.data_ptr(), xf.data_ptr(), yf.data_ptr()))
        assert ctx.last_kernel() == "fmm_kernel+argmin", ctx.last_kernel()
    torch.cuda.synchronize()
    assert int(torch.isnan(vol).sum()) == 0
    amax = float(exact.abs().max())
    err = (vol - exact).abs()
    bad = err > 1e-5 * exact.abs() + 1e-6 * amax
    assert not bool(bad.any()), "%d cells outside the tolerance, worst %g at cost %g" % (int(bad.sum()), float(err.max()), float(exact.reshape(-1)[err.argmax()]))
    # the arg-min form sees the same matrix-core costs: its index is the first minimum of `vol`
    assert torch.equal(idx, _first_min_index(vol) + 1)
    want = _first_min_index(exact)
    srt = torch.sort(exact.reshape(H1, W1, -1), dim=2).values
    near = (srt[..., 1] - srt[..., 0]) <= 2 * (1e-5 * srt[..., 1].abs() + 1e-6 * amax)
    differ = idx != want + 1
    assert not bool((differ & ~near).any()), "%d pixels differ from the exact first minimum outside ties" % int((differ & ~near).sum())
    assert int(differ.sum()) <= 0.01 * idx.numel() + 2, "differing pixels: %d of %d" % (int(differ.sum()), idx.numel())
    i0 = idx - 1
    lWin, tWin = (mw + 1) // 2 - 1, (mh + 1) // 2 - 1
    assert torch.equal(yf, (i0 // mw - tWin).to(torch.float32)) and torch.equal(xf, (i0 % mw - lWin).to(torch.float32))
    # the planted shift is what the interior finds
    if H1 > 20:
        assert float((idx[4:-4, 8:-8] == 3 * mw + 5 + 1).float().mean()) > 0.95
