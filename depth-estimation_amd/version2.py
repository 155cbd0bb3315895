"""version2/: the single-scale learned model (version2/network.lua, version2/test.lua).

    getNetwork(datap)          version2/network.lua:5-39    ParallelTable{filter1, filter2} -> SpatialMatching(hWin, wWin, false)
    getTrainerNetwork(datap)   version2/network.lua:41-47   ... -> Reshape(wWin * hWin) -> Minus -> LogSoftMax
    defaultDatap(...)          version2/test.lua:6-21       the script's `datap` table (lWin / tWin / rWin / bWin derived as there)
    network:getParameters()    -> flatParameters / loadParameters: `parameters:copy(torch.load('models/e106_no_bin'))` (test.lua:40-42)
    decodeFlow(output, datap)  version2/test.lua:45-51      min over the window, idx - 1, yflow / xflow
    flowPair(network, datap, prev, cur)                     test.lua:43-51 for one pair: staged through the modules, or ONE call
                                                            (dfe_version2_flow_pair_f32)

filter1 = Sequential{SpatialContrastiveNormalization(3, image.gaussian1D(normalization_k)), SpatialPadding(-lWin, -tWin, -rWin, -bWin),
conv layers}; filter2 = Sequential{the normalisation's clone, the conv layers' clones sharing weight / bias / gradWeight / gradBias}.
"""
import ctypes as C
import math

import torch

from ._lib import lib
from .context import get_ctx, ptr
from .network import (Sequential, ParallelTable, SpatialConvolution, SpatialContrastiveNormalization, gaussian1D, Minus, shared_clone)
from .nn import Module, SpatialMatching
from .radial import SpatialPadding


def _d(datap, key, default=None):
    return datap.get(key, default) if isinstance(datap, dict) else getattr(datap, key, default)


def defaultDatap(wImg=320, hImg=180, normalization_k=17, layers=((3, 17, 17, 32),), wWin=17, hWin=17):
    """The `datap` table of version2/test.lua:6-21 (layers[i] = {nIn, kW, kH, nOut} as network.lua:15-17 reads them)."""
    d = dict(wImg=wImg, hImg=hImg, normalization_k=normalization_k, layers=[list(l) for l in layers], wWin=wWin, hWin=hWin)
    d["hKernel"] = 1 + sum(l[2] - 1 for l in layers)
    d["wKernel"] = 1 + sum(l[1] - 1 for l in layers)
    d["lWin"] = math.ceil(wWin / 2) - 1
    d["tWin"] = math.ceil(hWin / 2) - 1
    d["rWin"] = wWin // 2
    d["bWin"] = hWin // 2
    return d


class Reshape(Module):
    """nn.Reshape(n): the trainer's patch-mode output 1 x 1 x hWin x wWin -> n (version2/network.lua:43)."""

    def __init__(self, *size):
        super().__init__()
        self.size = tuple(int(s) for s in size)

    def updateOutput(self, input):
        self.output = input.reshape(self.size)
        return self.output

    def updateGradInput(self, input, gradOutput):
        self.gradInput = gradOutput.reshape(input.shape)
        return self.gradInput


class LogSoftMax(Module):
    """nn.LogSoftMax over the last dimension (dfe_log_softmax_f32)."""

    def updateOutput(self, input):
        x = input.contiguous()
        N = x.shape[-1]
        out = torch.empty_like(x)
        ctx = get_ctx(x)
        ctx.check(lib().dfe_log_softmax_f32(ctx.handle, ptr(x), x.numel() // N, N, ptr(out)))
        self.output = out
        return out

    def updateGradInput(self, input, gradOutput):
        go = gradOutput.contiguous()
        N = go.shape[-1]
        gi = torch.empty_like(go)
        ctx = get_ctx(go)
        ctx.check(lib().dfe_log_softmax_backward_f32(ctx.handle, ptr(self.output), ptr(go), go.numel() // N, N, ptr(gi)))
        self.gradInput = gi
        return gi


def getNetwork(datap, device="cuda", generator=None):
    """version2/network.lua:5-39.  network.modules[0].modules = [filter1, filter2]; filter1.modules = [normalisation, SpatialPadding
    (crop), conv...]; filter2.modules = [normalisation, conv clones...]; network.modules[1] = SpatialMatching(hWin, wWin, false).
    The SpatialPadding takes its arguments in the order the script passes them -- (-lWin, -tWin, -rWin, -bWin) into
    nn.SpatialPadding(pad_l, pad_r, pad_t, pad_b) -- which is the intended crop only for windows with lWin == tWin and rWin == bWin
    (odd, square windows such as the script's 17 x 17); other windows fail in SpatialMatching's size check, as they would there."""
    layers = _d(datap, "layers")
    network = Sequential()
    filters = ParallelTable()
    network.add(filters)
    filter1 = Sequential()
    filters.add(filter1)
    filter1.add(SpatialContrastiveNormalization(layers[0][0], gaussian1D(_d(datap, "normalization_k"))))
    filter1.add(SpatialPadding(-_d(datap, "lWin"), -_d(datap, "tWin"), -_d(datap, "rWin"), -_d(datap, "bWin")))
    elems = [SpatialConvolution(l[0], l[3], l[1], l[2], device=device, generator=generator) for l in layers]
    filter2 = Sequential()
    filters.add(filter2)
    filter2.add(SpatialContrastiveNormalization(layers[0][0], gaussian1D(_d(datap, "normalization_k"))))
    for e in elems:
        filter1.add(e)
        filter2.add(shared_clone(e))
    network.add(SpatialMatching(_d(datap, "hWin"), _d(datap, "wWin"), False))
    network.getWeights = lambda: {"layer1": network.modules[0].modules[0].modules[2].weight}   # network.lua:32-36
    network.flatParameters = lambda: flatParameters(network)
    network.loadParameters = lambda flat: loadParameters(network, flat)
    return network


def getTrainerNetwork(datap, device="cuda", generator=None):
    """version2/network.lua:41-47 (patch mode: the matcher's output is 1 x 1 x hWin x wWin)."""
    network = getNetwork(datap, device=device, generator=generator)
    network.add(Reshape(_d(datap, "wWin") * _d(datap, "hWin")))
    network.add(Minus())
    network.add(LogSoftMax())
    return network


def _convs(network):
    return [m for m in network.modules[0].modules[0].modules if isinstance(m, SpatialConvolution)]


def flatParameters(network):
    """network:getParameters() (test.lua:40): one flat tensor of every learnable parameter in module order -- the first branch's
    convolution weights and biases (the second branch shares their storage; the normalisation modules hold no parameters of their own
    in nn's sense: Module:parameters() looks at self.weight / self.bias only)."""
    return torch.cat([t.reshape(-1) for m in _convs(network) for t in (m.weight, m.bias)])


def loadParameters(network, flat):
    """parameters:copy(torch.load(...)) (test.lua:41): in place, so the shared clones of the second branch follow."""
    flat = flat.reshape(-1)
    need = sum(m.weight.numel() + m.bias.numel() for m in _convs(network))
    if flat.numel() != need:
        raise ValueError("loadParameters: %d values for a network of %d parameters" % (flat.numel(), need))
    o = 0
    for m in _convs(network):
        for t in (m.weight, m.bias):
            t.copy_(flat[o : o + t.numel()].reshape(t.shape).to(t.device, t.dtype))
            o += t.numel()


def decodeFlow(output, datap):
    """version2/test.lua:45-51.  `output:min(3)` there is applied to the matcher's H x W x hWin x wWin output; the decode that follows
    (`floor(idx / wWin)`, `idx - yflow * wWin`) reads idx as the flat window index, i.e. the minimum over the whole window -- which is
    what is taken here (first minimum in index order; the script's literal min over dimension 3 alone would not give a window index).
    Returns (xflow, yflow) as int64 tensors, like the script's."""
    H, W = output.shape[0], output.shape[1]
    wWin = _d(datap, "wWin")
    idx = output.reshape(H, W, -1).argmin(dim=2)          # 0-based == torch7's idx:add(-1); argmin returns the first minimum
    yflow = torch.div(idx, wWin, rounding_mode="floor")
    xflow = idx - yflow * wWin - _d(datap, "lWin")
    yflow = yflow - _d(datap, "tWin")
    return xflow, yflow


def flowPair(network, datap, prev, cur, one_call=True, want_volume=False):
    """test.lua:43-51 for one frame pair: network:forward({prev, cur}) and the decode.  one_call: dfe_version2_flow_pair_f32 (normalisation,
    crop, convolutions, matcher, arg-min and decode on the device in one entry); else module by module.  Both give the same bits.
    Returns dict(xflow, yflow [float32 H1 x W1], index [int64, 1-based], volume or None)."""
    layers = _d(datap, "layers")
    hWin, wWin = _d(datap, "hWin"), _d(datap, "wWin")
    if not one_call:
        out = network.forward([prev, cur])
        H1, W1 = out.shape[0], out.shape[1]
        idx0 = out.reshape(H1, W1, -1).argmin(dim=2)
        xf, yf = decodeFlow(out, datap)
        return {"xflow": xf.to(torch.float32), "yflow": yf.to(torch.float32), "index": idx0 + 1, "volume": out if want_volume else None}
    from .multiscale import filter_layers_array

    f1 = network.modules[0].modules[0]
    scn = f1.modules[0]
    stack = Sequential()
    for m in f1.modules[2:]:
        stack.add(m)
    arr, nl, keep = filter_layers_array([stack])
    p, c = prev.contiguous(), cur.contiguous()
    if p.dtype != torch.float32 or c.dtype != torch.float32:
        raise TypeError("flowPair: float32 frames expected")
    Cc, H, W = p.shape
    hk = 1 + sum(l[2] - 1 for l in layers)
    wk = 1 + sum(l[1] - 1 for l in layers)
    H1, W1 = H - (hWin - 1) - (hk - 1), W - (wWin - 1) - (wk - 1)
    if H1 <= 0 or W1 <= 0:
        raise ValueError("flowPair: frame %dx%d too small" % (H, W))
    xf = torch.empty((H1, W1), dtype=torch.float32, device=p.device)
    yf = torch.empty_like(xf)
    idx = torch.empty((H1, W1), dtype=torch.int64, device=p.device)
    vol = torch.empty((H1, W1, hWin, wWin), dtype=torch.float32, device=p.device) if want_volume else None
    ctx = get_ctx(p)
    kp = scn.kernel.numpy().ctypes.data_as(C.POINTER(C.c_float))
    ctx.check(lib().dfe_version2_flow_pair_f32(ctx.handle, ptr(p), ptr(c), Cc, H, W, kp, scn.kernel.numel(), scn.threshold, scn.thresval, arr, nl,
                                               hWin, wWin, ptr(xf), ptr(yf), ptr(idx), ptr(vol)))
    return {"xflow": xf, "yflow": yf, "index": idx, "volume": vol}
