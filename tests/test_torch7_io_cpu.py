"""CPU suite, next-row N3: the Torch7 binary reader / writer (depth-estimation_amd/torch7_io.py) on the reference's own
calibration files (tests/golden/cal/*.cal are byte copies of /root/reference/radial/*.cal and
version2/rectified_gopro.cal -- data files, 538..664 bytes) and on round trips of weight-file shaped tables."""
import os
import struct

import numpy as np
import pytest

import depth_estimation_amd as d

CAL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cal")


def test_ardrone_calibration_fields_against_raw_bytes():
    path = os.path.join(CAL, "radial_ardrone.cal")
    raw = open(path, "rb").read()
    cal = d.load_calibration(path)
    # independent of the parser: the doubles / floats sit at fixed offsets of this 538-byte file
    assert struct.unpack_from("<d", raw, 28)[0] == 640.0 == cal["wImg"]          # after tag 3, index, n=6, key "wImg", tag 1
    assert struct.unpack_from("<d", raw, 52)[0] == 480.0 == cal["hImg"]
    k = raw.index(b"torch.FloatStorage") + len("torch.FloatStorage")
    assert struct.unpack_from("<q", raw, k)[0] == 9
    assert np.array_equal(np.frombuffer(raw, "<f4", 9, k + 8).reshape(3, 3), cal["K"])
    assert cal["K"].dtype == np.float32 and cal["K"][2].tolist() == [0.0, 0.0, 1.0]
    assert abs(cal["K"][0, 0] - 293.8247) < 1e-3 and abs(cal["K"][1, 2] - 251.6249) < 1e-3
    assert cal["distortion"].shape == (5,) and abs(cal["distortion"][0] + 0.37994) < 1e-6
    assert cal["bad_image_threshold"] == 0.2
    assert cal["sfm"] == {"max_points": 400.0, "points_quality": 0.001, "ransac_max_dist": 1.0}


@pytest.mark.parametrize("name,w,h,fx,rect", [
    ("radial_gopro", 1280, 720, 602.6632, None), ("radial_rectified_gopro", 1280, 720, 602.6632, None),
    ("version2_rectified_gopro", 1280, 720, 602.6632, True)])
def test_gopro_calibrations(name, w, h, fx, rect):
    cal = d.load_calibration(os.path.join(CAL, name + ".cal"))
    assert (cal["wImg"], cal["hImg"]) == (w, h) and abs(cal["K"][0, 0] - fx) < 1e-3
    assert cal.get("rectify") is rect
    if "rectified" in name:
        assert not cal["distortion"].any() and cal["sfm"][("trackerWinSize" if rect else "tracker_win_size")] == 21.0
    else:
        assert abs(cal["distortion"][1] - 0.142684) < 1e-6


def test_round_trip_weights_table():
    """The shape of a saveModel / saveNetwork file (opticalflow_model_io.lua:149-163): version, geometry table, list of
    weight tensors, strings, booleans."""
    rng = np.random.default_rng(0)
    obj = {
        "version": 9.0,
        "geometry": {"maxh": 8.0, "maxw": 8.0, "ratios": [1.0, 2.0, 4.0], "multiscale": True, "layers": [[3.0, 5.0, 5.0, 4.0], [4.0, 5.0, 5.0, 10.0]]},
        "weights": [rng.standard_normal((4, 3, 5, 5)).astype(np.float32), rng.standard_normal(4).astype(np.float32),
                    rng.integers(0, 9, (3, 2)).astype(np.int64), rng.standard_normal((2, 3)).astype(np.float64)],
        "model_descr": "nn.Sequential {...}",
        "score": None,
    }
    back = d.torch7_io.loads(d.torch7_io.dumps(obj))
    assert back["version"] == 9.0 and back["geometry"] == obj["geometry"] and back["model_descr"] == obj["model_descr"]
    assert "score" in back and back["score"] is None
    for a, b in zip(obj["weights"], back["weights"]):
        assert a.dtype == b.dtype and np.array_equal(a, b)


def test_strided_tensor_shared_storage_and_errors():
    # a 2x3 view (strides 1, 2: a transposed tensor) into a 6-element FloatStorage, written by hand
    st = np.arange(6, dtype=np.float32)
    w = struct.pack("<ii", 4, 1) + struct.pack("<i", 3) + b"V 1" + struct.pack("<i", 17) + b"torch.FloatTensor"
    w += struct.pack("<i", 2) + struct.pack("<qq", 2, 3) + struct.pack("<qq", 1, 2) + struct.pack("<q", 1)
    w += struct.pack("<ii", 4, 2) + struct.pack("<i", 3) + b"V 1" + struct.pack("<i", 18) + b"torch.FloatStorage" + struct.pack("<q", 6) + st.tobytes()
    t = d.torch7_io.loads(w)
    assert np.array_equal(t, st.reshape(3, 2).T)
    # the oldest files have no "V 1" marker
    old = struct.pack("<ii", 4, 1) + struct.pack("<i", 19) + b"torch.DoubleStorage" + struct.pack("<q", 2) + np.array([1.5, -2.0]).tobytes()
    assert d.torch7_io.loads(old).tolist() == [1.5, -2.0]
    # a table referenced twice is one object
    tw = struct.pack("<iii", 3, 1, 2) + struct.pack("<id", 1, 1.0) + struct.pack("<iii", 3, 2, 0) + struct.pack("<id", 1, 2.0) + struct.pack("<ii", 3, 2)
    two = d.torch7_io.loads(tw)
    assert isinstance(two, list) and two[0] is two[1]
    with pytest.raises(ValueError):
        d.torch7_io.loads(w[:-5])
    with pytest.raises(ValueError):
        d.torch7_io.loads(struct.pack("<i", 42))
    with pytest.raises(ValueError):
        d.load_calibration(os.path.join(CAL, "..", "golden_v1.npz"))


def test_malformed_tensor_geometry_is_rejected():
    """sizes, strides and the storage offset of a file are untrusted -- a tensor that would reach outside its storage,
    or has a negative size / stride / offset, raises instead of reading adjacent memory; tag 8 (the current Torch7's recursive
    function tag) is accepted like 6 / 7."""
    st = np.arange(6, dtype=np.float32)

    def tensor(size, stride, off):
        w = struct.pack("<ii", 4, 1) + struct.pack("<i", 3) + b"V 1" + struct.pack("<i", 17) + b"torch.FloatTensor"
        w += struct.pack("<i", len(size)) + b"".join(struct.pack("<q", x) for x in size) + b"".join(struct.pack("<q", x) for x in stride) + struct.pack("<q", off)
        w += struct.pack("<ii", 4, 2) + struct.pack("<i", 3) + b"V 1" + struct.pack("<i", 18) + b"torch.FloatStorage" + struct.pack("<q", 6) + st.tobytes()
        return w

    assert d.torch7_io.loads(tensor([2, 3], [3, 1], 1)).tolist() == [[0, 1, 2], [3, 4, 5]]
    assert d.torch7_io.loads(tensor([2, 2], [3, 1], 2)).tolist() == [[1, 2], [4, 5]]
    for size, stride, off in [([2, 4], [3, 1], 1), ([7], [1], 1), ([2, 3], [3, 1], 2), ([2, 3], [3, 1], 0), ([-1, 3], [3, 1], 1),
                              ([2, 3], [-3, 1], 4), ([2, 3], [1 << 40, 1], 1)]:
        with pytest.raises(ValueError):
            d.torch7_io.loads(tensor(size, stride, off))
    assert d.torch7_io.loads(tensor([0, 3], [3, 1], 1)).shape == (0, 3)
    with pytest.raises(ValueError):      # absurd dimension count
        d.torch7_io.loads(struct.pack("<ii", 4, 1) + struct.pack("<i", 17) + b"torch.FloatTensor" + struct.pack("<i", 1 << 20))
    for tag in (6, 7, 8):
        f = d.torch7_io.loads(struct.pack("<ii", tag, 1) + struct.pack("<i", 4) + b"\x1bLJ\x02" + struct.pack("<i", 0))
        assert isinstance(f, d.torch7_io.Torch7Function) and f.bytecode == b"\x1bLJ\x02" and f.upvalues is None


def _filter_params(filt):
    return [(m.weight, m.bias) for m in filt.modules if getattr(m, "weight", None) is not None]


@pytest.mark.parametrize("share", [True, False])
def test_save_load_model_round_trip_multiscale(tmp_path, share):
    """saveModel -> loadModel (opticalflow_model_io.lua:97-201) through a Torch7 file: a freshly constructed model gets the saved
    convolution weights under the reference's names, in place (shared clones follow); biases are not in the file -- the
    reference's getWeights lists weights only."""
    import torch

    geo = dict(maxh=8, maxw=8, ratios=[1, 2, 4], multiscale=True, layers=[[3, 5, 5, 4], [4, 5, 5, 4], [4, 5, 5, 10]], share_filters=share,
               hImg=96, wImg=128, output_extraction_method="max", maxhHR=32, maxwHR=32)
    model = d.getModelMultiscale(dict(geo), True, False, device="cpu", generator=torch.Generator().manual_seed(5))
    if not share:
        for i, f in enumerate(model.filters):
            for w, _ in _filter_params(f):
                w.mul_(1.0 + i)
    names = set(model.getWeights())
    assert names == ({"layer1", "layer2", "layer3"} if share else {"scale%d_layer%d" % (r, i) for r in (1, 2, 4) for i in (1, 2, 3)})
    path = str(tmp_path / "model_e000010")
    d.saveModel(path, geo, {"rate": 0.01, "num_images": 10}, model)
    raw = d.torch7_io.load(path)
    assert raw["version"] == 9 and set(raw["weights"]) == names and raw["geometry"]["ratios"] == [1.0, 2.0, 4.0]
    ret = d.loadModel(path, True, False, wImg=640, hImg=480, device="cpu")
    g2 = ret["geometry"]
    assert g2["wImg"] == 640 and g2["hImg"] == 480 and g2["training_mode"] is False and g2["ratios"] == [1, 2, 4] and g2["layers"][2] == [4, 5, 5, 10]
    assert g2["hKernel"] == 13 and isinstance(ret["model"], d.MultiscaleModel)
    for k, w in ret["model"].getWeights().items():
        assert torch.equal(w, model.getWeights()[k])
    if share:      # every scale computes with the ONE loaded tensor
        assert ret["model"].filters[2].modules[0].weight is ret["model"].filters[0].modules[0].weight
    # prefilter = true: the weights land in the stand-alone filter, the model itself has none (:184-195)
    pre = d.loadModel(path, True, True, device="cpu")
    assert pre["model"].prefiltered and pre["model"].getWeights() == {}
    for k, w in pre["filter"].getWeights().items():
        assert torch.equal(w, model.getWeights()[k])
    # loadWeightsFrom: names the model does not have are skipped
    other = d.getModelMultiscale(dict(geo), True, False, device="cpu", generator=torch.Generator().manual_seed(9))
    assert not torch.equal(next(iter(other.getWeights().values())), next(iter(model.getWeights().values())))
    d.loadWeightsFrom(other, path)
    for k, w in other.getWeights().items():
        assert torch.equal(w, model.getWeights()[k])
    d.torch7_io.save(path, dict(raw, version=8.0))
    with pytest.raises(ValueError, match="version 9"):
        d.loadModel(path, True, False, device="cpu")


def test_save_load_model_single_scale_and_shape_mismatch(tmp_path):
    import torch

    geo = dict(maxh=16, maxw=16, multiscale=False, layers=[[3, 5, 5, 8], [8, 5, 5, 10]], hImg=180, wImg=320, output_extraction_method="max")
    model = d.getModel(geo, True, False, device="cpu", generator=torch.Generator().manual_seed(2))
    path = str(tmp_path / "m")
    d.saveModel(path, geo, None, model)
    ret = d.loadModel(path, False, False, device="cpu")
    assert ret["geometry"]["training_mode"] is True and type(ret["model"].modules[-1]).__name__ == "Log2"
    assert torch.equal(ret["model"].getWeights()["layer2"], model.getWeights()["layer2"])
    assert ret["model"].modules[0].modules[1].modules[0].weight is ret["model"].modules[0].modules[0].modules[0].weight
    raw = d.torch7_io.load(path)
    raw["weights"]["layer1"] = raw["weights"]["layer1"][:, :2]
    d.torch7_io.save(path, raw)
    with pytest.raises(ValueError, match="layer1"):
        d.loadModel(path, True, False, device="cpu")


def test_save_load_radial_network(tmp_path):
    """saveNetwork / loadTesterNetwork / loadTrainerNetwork (radial/radial_opticalflow_network.lua:120-154): version 1, networkp,
    weights = {weights list, bias list}; both filter branches of the loaded network share the loaded tensors."""
    import torch

    networkp = dict(hImg=180, wImg=320, hInput=120, wInput=136, hWin=15, layers=[[3, 1, 17, 5], "tanh", [5, 17, 1, 10]])
    net = d.getTesterNetwork(networkp, device="cpu", generator=torch.Generator().manual_seed(4))
    fn = d.saveNetwork(str(tmp_path), 7, networkp, net)
    assert fn.endswith("model_7")
    raw = d.torch7_io.load(fn)
    assert raw["version"] == 1 and len(raw["weights"]) == 2 and len(raw["weights"][0]) == 2 and raw["networkp"]["layers"][1] == "tanh"
    net2, np2 = d.loadTesterNetwork(fn, device="cpu")
    assert np2["layers"] == [[3, 1, 17, 5], "tanh", [5, 17, 1, 10]] and np2["hWin"] == 15
    (w1, w2), (b1, b2) = d.model_io.getWeights(net)
    (v1, v2), (c1, c2) = d.model_io.getWeights(net2)
    assert torch.equal(w1, v1) and torch.equal(w2, v2) and torch.equal(b1, c1) and torch.equal(b2, c2)
    assert net2.modules[0].modules[0].modules[1].modules[0].weight is v1       # branch 1's filter holds the same tensor as the clone
    net3, _ = d.loadTrainerNetwork(fn, device="cpu")
    assert type(net3.modules[-1]).__name__ == "LogSoftMaxRows" and torch.equal(d.model_io.getWeights(net3)[0][1], w2)
    d.torch7_io.save(fn, dict(raw, version=2.0))
    with pytest.raises(ValueError, match="version"):
        d.loadTesterNetwork(fn, device="cpu")
