"""Trained-weight files of the reference, onto the modules of this package (next-row N3, second half).

* `saveModel` / `loadModel` / `loadWeightsFrom` -- opticalflow_model_io.lua:97-207: a Torch7 table {version = 9, weights =
  model:getWeights(), geometry, learning, model_descr, score, + four Lua closures}.  `weights` maps 'layer<i>' (single scale or
  shared filters) / 'scale<r>_layer<i>' (one stack per scale) [/ 'cascad'] to the CONVOLUTION WEIGHTS only -- the reference's
  getWeights never lists the biases (opticalflow_model.lua:66-76), so a loaded model keeps the biases it was constructed with.
* `saveNetwork` / `loadTesterNetwork` / `loadTrainerNetwork` / `getWeights` / `copyWeights` --
  radial/radial_opticalflow_network.lua:76-156: {version = 1, networkp, weights = {list of weights, list of biases}}.

Files are read and written with torch7_io (no Torch7).  The closures of a reference-written file (getModel, getFilter, ...) come
back as opaque Torch7Function objects: which constructor to call is decided from geometry.multiscale instead, which is what
the saved closure was (opticalflow_model_io.lua:151-155).  Files written here carry no closures."""
import os

import numpy as np
import torch

from . import torch7_io


def _plain(o):
    """Lua numbers come back as floats: integral ones to int, recursively (geometry / networkp tables)."""
    if isinstance(o, float) and o == int(o) and abs(o) < 2 ** 53:
        return int(o)
    if isinstance(o, list):
        return [_plain(v) for v in o]
    if isinstance(o, dict):
        return {k: _plain(v) for k, v in o.items()}
    return o


def _to_file(o):
    if isinstance(o, torch.Tensor):
        return o.detach().cpu().numpy()
    if isinstance(o, dict):
        return {k: _to_file(v) for k, v in o.items() if not callable(v)}
    if isinstance(o, (list, tuple)):
        return [_to_file(v) for v in o]
    return o


def _copy_into(dst, src, what):
    src = torch.from_numpy(np.ascontiguousarray(src)) if isinstance(src, np.ndarray) else src
    if tuple(dst.shape) != tuple(src.shape):
        raise ValueError("%s: file has %s, the model %s" % (what, tuple(src.shape), tuple(dst.shape)))
    dst.copy_(src.to(dst.dtype))          # weights[k]:copy(loaded.weights[k]): IN PLACE, so shared clones follow


# ---- opticalflow_model_io.lua ---------------------------------------------------------------------------------------
def saveModel(path, geometry, learning, model, score=None):
    """opticalflow_model_io.lua:97-163 (the table; the reference derives a directory name from the hyper-parameters, :99-146 --
    here the caller names the file)."""
    tosave = {"version": 9, "model_descr": repr(type(model).__name__), "weights": _to_file(model.getWeights()),
              "geometry": _to_file(dict(geometry)), "learning": _to_file(dict(learning or {})), "score": _to_file(score)}
    torch7_io.save(path, tosave)


def loadModel(filename, full_output=True, prefilter=False, wImg=None, hImg=None, device="cuda"):
    """opticalflow_model_io.lua:166-201 -> dict(geometry, model, [filter], score)."""
    from .multiscale import getModelMultiscale, getMultiscalePrefilter
    from .network import getFilter, getModel

    loaded = torch7_io.load(filename)
    if not isinstance(loaded, dict) or "version" not in loaded:
        raise ValueError("%s is not a saveModel file" % filename)
    if loaded["version"] < 9:
        raise ValueError("loadModel: can't load before version 9 (structure has changed too much)")   # :171
    geometry = _plain(loaded["geometry"])
    if wImg:
        geometry["wImg"] = int(wImg)
    if hImg:
        geometry["hImg"] = int(hImg)
    geometry["training_mode"] = not full_output                                                         # :176-180
    make = getModelMultiscale if geometry.get("multiscale") else getModel
    ret = {"geometry": geometry, "score": loaded.get("score"), "getKernels": loaded.get("getKernels")}
    ret["model"] = make(geometry, full_output, prefilter, device=device)
    weights = loaded.get("weights") or {}

    def _w(k):
        if k not in weights:
            raise ValueError("loadModel: %s holds no weight named '%s' (it has %s)" % (filename, k, sorted(weights)))
        return weights[k]

    if prefilter:                                                                                       # :184-195
        filt = getFilter(geometry, device=device)
        ret["filter"] = getMultiscalePrefilter(geometry, filt) if geometry.get("multiscale") else filt
        for k, w in ret["filter"].getWeights().items():
            _copy_into(w, _w(k), k)
    for k, w in ret["model"].getWeights().items():                                                      # :192-199
        _copy_into(w, _w(k), k)
    return ret


def loadWeightsFrom(model, filename):
    """opticalflow_model_io.lua:203-214: copies the file's weights onto the model's parameters of the same name; names the
    model does not have are skipped."""
    loaded = torch7_io.load(filename)
    if loaded["version"] < 9:
        raise ValueError("Can't load weights from file before version 9")
    mine = model.getWeights()
    for k, v in (loaded.get("weights") or {}).items():
        if k in mine:
            _copy_into(mine[k], v, k)


# ---- radial/radial_opticalflow_network.lua ---------------------------------------------------------------------------
CURRENT_VERSION = 1   # :120


def getWeights(network):
    """:76-90 -> (weights, biases) of network.modules[1].modules[2].modules (the shared clone: the same tensors as branch 1's)."""
    layers = network.modules[0].modules[1].modules
    ws = [m.weight for m in layers if getattr(m, "weight", None) is not None]
    bs = [m.bias for m in layers if getattr(m, "bias", None) is not None]
    return ws, bs


def copyWeights(srcnetwork, dstnetwork):
    """:92-109: src is a network or a {weights, biases} pair."""
    if isinstance(srcnetwork, (list, tuple)) and len(srcnetwork) == 2 and isinstance(srcnetwork[0], (list, tuple)):
        sw, sb = srcnetwork
    else:
        sw, sb = getWeights(srcnetwork)
    dw, db = getWeights(dstnetwork)
    assert len(sw) == len(dw) and len(sb) == len(db)
    for i, (d_, s_) in enumerate(zip(dw, sw)):
        _copy_into(d_, s_, "weights[%d]" % (i + 1))
    for i, (d_, s_) in enumerate(zip(db, sb)):
        _copy_into(d_, s_, "bias[%d]" % (i + 1))


def saveNetwork(dir, iEpoch, networkp, network):
    """:121-130 -> the file name it wrote (dir/model_<iEpoch>)."""
    filename = os.path.join(dir, "model_%s" % iEpoch)
    ws, bs = getWeights(network)
    torch7_io.save(filename, {"version": CURRENT_VERSION, "networkp": _to_file(dict(networkp)), "weights": [_to_file(ws), _to_file(bs)]})
    return filename


def checkVersion(loaded):
    """:132-136"""
    if loaded["version"] != CURRENT_VERSION:
        raise ValueError("Input file has version %s but is required to have version %s" % (loaded["version"], CURRENT_VERSION))


def _load_network(filename, make, device):
    loaded = torch7_io.load(filename)
    checkVersion(loaded)
    networkp = _plain(loaded["networkp"])
    network = make(networkp, device=device)
    copyWeights(loaded["weights"], network)
    return network, networkp


def loadTrainerNetwork(filename, device="cuda"):
    """:138-145"""
    from .radial import getTrainerNetwork

    return _load_network(filename, getTrainerNetwork, device)


def loadTesterNetwork(filename, device="cuda"):
    """:147-154"""
    from .radial import getTesterNetwork

    return _load_network(filename, getTesterNetwork, device)
