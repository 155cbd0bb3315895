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
