"""Multiscale matcher with the reference's names (opticalflow_model_multiscale.lua, CascadingAddTable.lua)
for the raw-patch (identity filter) case: pyramid volumes -> softmin -> cascade -> ring extraction."""
import ctypes as C

import math

import torch

from ._lib import lib, ratios_array
from .context import get_ctx, ptr
from .nn import Module, _f32c
from .opticalflow_model import _g


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class CascadingAddTable(Module):
    """nn.CascadingAddTable(ratios, trainable, single_beta) -- CascadingAddTable.lua:7-135 (forward).
    HEAD sums plainly: the Mul2 gains and the Power normaliser are commented out (:29,46,61), so
    `trainable` / `single_beta` do not change updateOutput."""

    def __init__(self, ratios, trainable=True, single_beta=False):
        super().__init__()
        self.ratios = [int(r) for r in ratios]
        self.trainable = trainable

    def updateOutput(self, input):
        for t in input:
            if t.dim() != 3:
                raise ValueError("nn.CascadingAddTable: input must be a table of 3D-tensors (HxW) x Kh x Kw")  # :109-112
        if len(input) != len(self.ratios):
            raise ValueError("nn.CascadingAddTable: input and ratios must have the same size")  # :114-116
        ins = [t.contiguous() for t in input]
        P, maxh, maxw = ins[0].shape
        for t in ins:
            if t.dtype != torch.float32 or tuple(t.shape) != (P, maxh, maxw):
                raise ValueError("nn.CascadingAddTable: inputs must be FloatTensors of one size")
        outs = [torch.empty_like(t) for t in ins]
        ctx = get_ctx(ins[0])
        r, n = ratios_array(self.ratios)
        ctx.check(lib().dfe_cascading_add_f32(ctx.handle, _ptr_array(ins), r, n, P, maxh, maxw, _ptr_array(outs)))
        self.output = outs
        return outs

    def updateGradInput(self, input, gradOutput):
        """CascadingAddTable.lua:137-154.  HEAD's graph holds no trainable parameter (Mul2 / Power are commented
        out), so accGradParameters has nothing to accumulate and `backward` is this."""
        if len(gradOutput) != len(self.ratios):
            raise ValueError("nn.CascadingAddTable: input and ratios must have the same size")
        gos = [t.contiguous() for t in gradOutput]
        P, maxh, maxw = gos[0].shape
        for t in gos:
            if t.dtype != torch.float32 or tuple(t.shape) != (P, maxh, maxw):
                raise ValueError("nn.CascadingAddTable: gradOutputs must be FloatTensors of one size")
        gis = [torch.empty_like(t) for t in gos]
        ctx = get_ctx(gos[0])
        r, n = ratios_array(self.ratios)
        ctx.check(lib().dfe_cascading_add_backward_f32(ctx.handle, _ptr_array(gos), r, n, P, maxh, maxw, _ptr_array(gis)))
        self.gradInput = gis
        return gis

    def accGradParameters(self, input, gradOutput, scale=1.0):   # :156-164, nothing trainable at HEAD
        return None

    def backward(self, input, gradOutput, scale=1.0):
        return self.updateGradInput(input, gradOutput)


def _pad_to_multiple(i0, i1, rmax):
    """opticalflow_model_multiscale.lua:234-248: nn.SpatialPadding(0,0,0,0) with pad_b / pad_r up to the next multiple of the
    coarsest ratio, zeros; the output keeps the padded size (nothing crops it back)."""
    Cc, H, W = i0.shape
    if H % rmax == 0 and W % rmax == 0:
        return i0, i1, H, W
    th, tw = rmax * math.ceil(H / rmax), rmax * math.ceil(W / rmax)
    p0 = torch.zeros((Cc, th, tw), dtype=i0.dtype, device=i0.device)
    p1 = torch.zeros_like(p0)
    p0[:, :H, :W] = i0
    p1[:, :H, :W] = i1
    return p0, p1, th, tw


def filter_layers_array(filters):
    """ctypes dfe_filter_layer[len(filters)][nlayers] for a list of getFilter(geometry) stacks (one per scale, or one shared);
    returns (array, nlayers, keepalive)."""
    from ._lib import FilterLayer
    from .network import SpatialConvolution, SpatialConvolutionMap, Tanh

    rows, keep = [], []
    for filt in filters:
        row = []
        for m in filt.modules:
            if isinstance(m, Tanh):
                if not row:
                    raise ValueError("filter stack starts with nn.Tanh")
                row[-1].tanh_after = 1
                continue
            L = FilterLayer()
            w, b = m.weight.contiguous(), (m.bias.contiguous() if m.bias is not None else None)
            keep += [w, b]
            L.kH, L.kW, L.nOut = m.kH, m.kW, m.nOutputPlane
            L.weight, L.bias = w.data_ptr(), (b.data_ptr() if b is not None else None)
            if isinstance(m, SpatialConvolutionMap):
                L.nIn, L.conn, L.nConn = m.nInputPlane, m._conn_dev.data_ptr(), m.connTable.shape[0]
                keep.append(m._conn_dev)
            elif isinstance(m, SpatialConvolution):
                L.nIn, L.conn, L.nConn = m.nInputPlane, None, 0
            else:
                raise ValueError("filter stack: unsupported module %s" % type(m).__name__)
            L.tanh_after = 0
            row.append(L)
        rows.append(row)
    nl = len(rows[0])
    if any(len(r) != nl for r in rows):
        raise ValueError("the scales' filter stacks must have the same number of layers")
    arr = (FilterLayer * (len(rows) * nl))(*[L for r in rows for L in r])
    return arr, nl, keep


class MultiscalePrefilter(Module):
    """getMultiscalePrefilter(geometry, filter) -- opticalflow_model_multiscale.lua:134-173: an nn.ConcatTable with, per ratio,
    SpatialDownSampling(r, r) -> SpatialZeroPadding(hPatch2-1 / wPatch2-1 split floor / ceil) -> the filter (sharing the first
    one's parameters when geometry.share_filters, an independent copy otherwise).  forward(frame C x H x W) -> list over the
    ratios of K x (H/r + maxh-1) x (W/r + maxw-1) feature maps."""

    def __init__(self, geometry, filt):
        from .network import _SharedFilter
        import copy

        super().__init__()
        self.geometry = geometry
        self.ratios = [int(r) for r in _g(geometry, "ratios")]
        self.share = bool(_g(geometry, "share_filters", False))
        self.modules = []
        for i, r in enumerate(self.ratios):
            if i == 0:
                self.modules.append(filt)
            elif self.share:
                self.modules.append(_SharedFilter(filt))         # filter:clone('weight', 'bias', 'gradWeight', 'gradBias')
            else:
                f = copy.deepcopy(filt)                          # filter:clone(): its own parameters, initialised equal
                f.getWeights = (lambda f=f: __import__("depth_estimation_amd").network.filter_weights(f))
                self.modules.append(f)

    def getWeights(self):
        """:156-170: the first stack's names when shared, else 'scale<r>_layer<i>'."""
        from .network import filter_weights

        if self.share:
            return filter_weights(self.modules[0])
        out = {}
        for r, f in zip(self.ratios, self.modules):
            for n, w in filter_weights(f).items():
                out["scale%d_%s" % (r, n)] = w
        return out

    def updateOutput(self, input):
        g = self.geometry
        x = _f32c(input, "MultiscalePrefilter: input")
        Cc, H, W = x.shape
        hp, wp = _g(g, "hPatch2") - 1, _g(g, "wPatch2") - 1
        ctx = get_ctx(x)
        outs = []
        for r, f in zip(self.ratios, self.modules):
            d = x
            if r > 1:
                d = torch.empty((Cc, H // r, W // r), dtype=torch.float32, device=x.device)
                ctx.check(lib().dfe_downsample_box_f32(ctx.handle, ptr(x), Cc, H, W, r, ptr(d)))
            d = torch.nn.functional.pad(d, (wp // 2, wp - wp // 2, hp // 2, hp - hp // 2))      # zero padding: plumbing
            outs.append(f.forward(d.contiguous()))
        self.output = outs
        return outs


class MultiscaleModel(Module):
    """What getModelMultiscale(geometry, full_image, prefiltered):forward(input) computes (opticalflow_model_multiscale.lua:175-333):
    the hImg x wImg x nclasses tensor that processOutput consumes (log of it, 1 x 1 x nclasses, in training mode with a focus).

    * filters = None: the raw-patch (identity) filter, input {I0, I1}.
    * filters = a getFilter(geometry) stack per scale (one shared stack when geometry.share_filters): the learned matcher,
      input {I0, I1} (:196-211, 219-226).
    * prefiltered = True (:190-191, 257-264): input is a list over the ratios of [feat0_s, feat1_s], the feature maps
      getMultiscalePrefilter makes of the two frames; frame 0's are cropped by the search window here (filter1's
      SpatialZeroPadding with negative pads, :198-202).
    Keeps the per-scale cost volumes (`self.volumes`, native scale) and probabilities (`self.probs`)."""

    def __init__(self, geometry, filters=None, prefiltered=False):
        super().__init__()
        self.geometry = geometry
        g = geometry
        ratios = [int(r) for r in _g(g, "ratios")]
        assert ratios[0] == 1  # :182
        rmax = ratios[-1]
        for r in ratios:  # :183-188
            k = rmax - r
            assert (_g(g, "maxh") * k) % 2 == 0 and (_g(g, "maxw") * k) % 2 == 0
        self.ratios = ratios
        self.filters = filters            # list over the scales (entries may share parameters), or None
        self.prefiltered = bool(prefiltered)
        self.cascad = CascadingAddTable(ratios, _g(g, "cascad_trainable_weights", False), _g(g, "single_beta", False))
        self._focus = None

    # ---- the reference's model:focus(x, y) (:339-345): evaluate the single pixel (x, y) (1-based, as in Lua) -- training mode
    def focus(self, x=None, y=None):
        self._focus = None if x is None else (int(x), int(y))

    def getWeights(self):
        """:347-370: filter weights as 'layer<i>' (shared) or 'scale<r>_layer<i>'; 'cascad' only with trainable cascade weights,
        which HEAD's CascadingAddTable does not have (Mul2 commented out)."""
        from .network import filter_weights

        out = {}
        if self.filters is None or self.prefiltered:
            return out
        if _g(self.geometry, "share_filters", False):
            return filter_weights(self.filters[0])
        for r, f in zip(self.ratios, self.filters):
            for n, w in filter_weights(f).items():
                out["scale%d_%s" % (r, n)] = w
        return out

    def _hk(self):
        g = self.geometry
        return _g(g, "hKernel"), _g(g, "wKernel")

    def _scale_features(self, i0, i1, H, W):
        """per scale: (feat0 K x Hs x Ws, feat1 K x (Hs+maxh-1) x (Ws+maxw-1)) through the staged C calls."""
        g = self.geometry
        maxh, maxw = _g(g, "maxh"), _g(g, "maxw")
        kh, kw = self._hk()
        hp, wp = maxh - 1 + kh - 1, maxw - 1 + kw - 1
        ct, cl = (maxh - 1) // 2, (maxw - 1) // 2
        ctx = get_ctx(i0)
        Cc = i0.shape[0]
        feats = []
        for s, r in enumerate(self.ratios):
            fr = []
            for img in (i0, i1):
                d = img
                if r > 1:
                    d = torch.empty((Cc, H // r, W // r), dtype=torch.float32, device=img.device)
                    ctx.check(lib().dfe_downsample_box_f32(ctx.handle, ptr(img), Cc, H, W, r, ptr(d)))
                fr.append(torch.nn.functional.pad(d, (wp // 2, wp - wp // 2, hp // 2, hp - hp // 2)))
            a = fr[0][:, ct : fr[0].shape[1] - (maxh - 1 - ct), cl : fr[0].shape[2] - (maxw - 1 - cl)].contiguous()   # filter1's crop (:198-202)
            f = self.filters[s]
            f0 = f.forward(a)
            f0 = f0.clone() if f0 is f.output else f0
            f1 = f.forward(fr[1].contiguous())
            feats.append((f0, f1))
        return feats

    def _volumes(self, input, f16_scale=None):
        """-> (volumes per scale [Hs][Ws][maxh][maxw], H, W): raw frames, learned filters or prefiltered features."""
        from .nn import SpatialMatching

        g = self.geometry
        maxh, maxw = _g(g, "maxh"), _g(g, "maxw")
        l = lib()
        vols = []
        if self.prefiltered:
            ct, cl = (maxh - 1) // 2, (maxw - 1) // 2
            H = W = None
            for s, (f0, f1) in enumerate(input):
                f0, f1 = _f32c(f0, "MultiscaleModel: features"), _f32c(f1, "MultiscaleModel: features")
                a = f0[:, ct : f0.shape[1] - (maxh - 1 - ct), cl : f0.shape[2] - (maxw - 1 - cl)].contiguous()
                vols.append(SpatialMatching(maxh, maxw).forward([a, f1]))
                if s == 0:
                    H, W = a.shape[1], a.shape[2]
        else:
            i0, i1 = input
            i0, i1 = _f32c(i0, "MultiscaleModel: input[1]"), _f32c(i1, "MultiscaleModel: input[2]")   # the C ABI is typed: float32 frames
            i0, i1, H, W = _pad_to_multiple(i0, i1, self.ratios[-1])
            Cc = i0.shape[0]
            ctx = get_ctx(i0)
            if self.filters is None:
                kh, kw = self._hk()
                for r in self.ratios:
                    vol = torch.empty((H // r, W // r, maxh, maxw), dtype=torch.float32, device=i0.device)
                    ctx.check(l.dfe_pyramid_scale_volume_f32(ctx.handle, ptr(i0), ptr(i1), Cc, H, W, r, kh, kw, maxh, maxw, ptr(vol)))
                    vols.append(vol)
            else:
                for f0, f1 in self._scale_features(i0, i1, H, W):
                    vols.append(SpatialMatching(maxh, maxw).forward([f0, f1]))
        if f16_scale:   # what an fp16 volume holds: half(cost * scale) read back as float(stored) * (1 / scale)
            vols = [(v * float(f16_scale)).to(torch.float16).to(torch.float32) * (1.0 / float(f16_scale)) for v in vols]
        return vols, H, W

    def _probs(self, vols):
        N = vols[0].shape[2] * vols[0].shape[3]
        ctx = get_ctx(vols[0])
        probs = []
        for vol in vols:
            prob = torch.empty_like(vol)
            ctx.check(lib().dfe_softmin_f32(ctx.handle, ptr(vol), vol.numel() // N, N, ptr(prob)))
            probs.append(prob)
        return probs

    def updateOutput(self, input):
        g = self.geometry
        maxh, maxw = _g(g, "maxh"), _g(g, "maxw")
        if self._focus is not None:
            return self._forward_focused(input)
        self.volumes, H, W = self._volumes(input)
        self.probs = self._probs(self.volumes)
        ctx = get_ctx(self.volumes[0])
        rr, n = ratios_array(self.ratios)
        ncls = lib().dfe_multi_nclasses(maxh, maxw, rr, n)
        out = torch.empty((H, W, ncls), dtype=torch.float32, device=self.volumes[0].device)
        ctx.check(lib().dfe_cascade_ring_f32(ctx.handle, _ptr_array(self.probs), rr, n, H, W, maxh, maxw, ptr(out)))
        self.output = out
        return out

    def _forward_focused(self, input):
        """training mode with model:focus(x, y): the class vector of the one pixel (x, y) -- 1 x 1 x nclasses, through nn.Log2(1e-10)
        when geometry.training_mode (:326-337).  Computed from per-scale crops that hold exactly that pixel's receptive field
        (what nnx's focused SpatialPyramid evaluates), with the same kernels as the full frame: equal to [y-1][x-1] of the full
        output."""
        from .glue import Log2
        from .nn import SpatialMatching

        g = self.geometry
        if self.prefiltered:
            raise NotImplementedError("focus on prefiltered inputs")
        maxh, maxw = _g(g, "maxh"), _g(g, "maxw")
        kh, kw = self._hk()
        hp, wp = maxh - 1 + kh - 1, maxw - 1 + kw - 1
        ct, cl = (maxh - 1) // 2, (maxw - 1) // 2
        x, y = self._focus[0] - 1, self._focus[1] - 1
        i0, i1 = input
        i0, i1 = _f32c(i0, "MultiscaleModel: input[1]"), _f32c(i1, "MultiscaleModel: input[2]")
        i0, i1, H, W = _pad_to_multiple(i0, i1, self.ratios[-1])
        if not (0 <= x < W and 0 <= y < H):
            raise ValueError("focus (%d, %d) outside the %d x %d frame" % (x + 1, y + 1, W, H))
        Cc = i0.shape[0]
        ctx = get_ctx(i0)
        probs = []
        for s, r in enumerate(self.ratios):
            ys, xs = y // r, x // r
            crops = []
            for img in (i0, i1):
                # the (hp+1) x (wp+1) window of the padded down-sampled frame around scale pixel (ys, xs): rows ys .. ys+hp
                ry0, rx0 = ys - hp // 2, xs - wp // 2
                win = torch.zeros((Cc, hp + 1, wp + 1), dtype=torch.float32, device=img.device)
                sy0, sy1 = max(ry0, 0), min(ry0 + hp + 1, H // r)
                sx0, sx1 = max(rx0, 0), min(rx0 + wp + 1, W // r)
                if sy1 > sy0 and sx1 > sx0:
                    src = img[:, sy0 * r : sy1 * r, sx0 * r : sx1 * r].contiguous()
                    d = src
                    if r > 1:
                        d = torch.empty((Cc, sy1 - sy0, sx1 - sx0), dtype=torch.float32, device=img.device)
                        ctx.check(lib().dfe_downsample_box_f32(ctx.handle, ptr(src), Cc, src.shape[1], src.shape[2], r, ptr(d)))
                    win[:, sy0 - ry0 : sy1 - ry0, sx0 - rx0 : sx1 - rx0] = d
                crops.append(win)
            a = crops[0][:, ct : ct + kh, cl : cl + kw].contiguous()
            b = crops[1]
            if self.filters is not None:
                f = self.filters[s]
                fa = f.forward(a)
                fa = fa.clone() if fa is f.output else fa
                fb = f.forward(b)
                vol = SpatialMatching(maxh, maxw).forward([fa, fb])
            else:
                # raw patches: both (hp+1) x (wp+1) crops through the frame op, whose own crop of frame 0 is filter1's (one output pixel)
                vol = torch.empty((1, 1, maxh, maxw), dtype=torch.float32, device=i0.device)
                c0, c1 = crops[0].contiguous(), crops[1].contiguous()
                ctx.check(lib().dfe_ssd_cost_volume_f32(ctx.handle, ptr(c0), ptr(c1), Cc, hp + 1, wp + 1, kh, kw, maxh, maxw, ptr(vol)))
            probs.append(self._probs([vol])[0])
        rr, n = ratios_array(self.ratios)
        ncls = lib().dfe_multi_nclasses(maxh, maxw, rr, n)
        # the cascade of a single pixel: every scale's window is "the" pixel's ([P = 1][maxh][maxw] per scale)
        outs = self.cascad.forward([p.reshape(1, maxh, maxw) for p in probs])
        out = _ring_join(outs, self.ratios, maxh, maxw).reshape(1, 1, ncls)
        if _g(g, "training_mode", False):
            out = Log2(1e-10).forward(out)
        self.output = out
        return out

    def forwardFlow(self, input, process_full=True, one_call=True, f16_scale=None):
        """model:forward(input) followed by processOutput(geometry, output, process_full) for the 'max' extraction
        without a threshold (opticalflow_model.lua:201-252), fused: the H x W x nclasses tensor is never built
        (dfe_cascade_flow_f32).  Returns the same table: index, confidences (all 1), y, x [, full, full_confidences].
        one_call: everything inside ONE C call (dfe_multiscale_flow_pair_f32 / _f16 / _filtered_f32).
        f16_scale: the per-scale cost volumes are stored as half(cost * f16_scale) (the staged path rounds its fp32 volumes to
        half precision the same way)."""
        g = self.geometry
        maxh, maxw = _g(g, "maxh"), _g(g, "maxw")
        kh, kw = self._hk()
        l = lib()
        N = maxh * maxw
        rr, n = ratios_array(self.ratios)
        if one_call and not self.prefiltered and (self.filters is not None or kh == kw):
            i0, i1 = input
            i0, i1 = _f32c(i0, "MultiscaleModel: input[1]"), _f32c(i1, "MultiscaleModel: input[2]")
            i0, i1, H, W = _pad_to_multiple(i0, i1, self.ratios[-1])
            Cc = i0.shape[0]
            ctx = get_ctx(i0)
            idx = torch.empty((H, W), dtype=torch.int64, device=i0.device)
            flow = torch.empty((2, H, W), dtype=torch.float32, device=i0.device)
            if self.filters is not None:
                share = bool(_g(g, "share_filters", False))
                arr, nl, keep = filter_layers_array(self.filters[:1] if share else self.filters)
                ctx.check(l.dfe_multiscale_flow_pair_filtered_f32(ctx.handle, ptr(i0), ptr(i1), Cc, H, W, maxh, maxw, rr, n, arr, nl, 1 if share else 0,
                                                                  float(f16_scale or 0.0), ptr(flow), ptr(idx)))
                del keep
            elif f16_scale:
                ctx.check(l.dfe_multiscale_flow_pair_f16(ctx.handle, ptr(i0), ptr(i1), Cc, H, W, kh, maxh, maxw, rr, n, float(f16_scale), ptr(flow), ptr(idx)))
            else:
                ctx.check(l.dfe_multiscale_flow_pair_f32(ctx.handle, ptr(i0), ptr(i1), Cc, H, W, kh, maxh, maxw, rr, n, ptr(flow), ptr(idx)))
            fy, fx = flow[0], flow[1]
            self.volumes, self.probs = None, None
        else:
            self.volumes, H, W = self._volumes(input, f16_scale)
            self.probs = self._probs(self.volumes)
            dev = self.volumes[0].device
            ctx = get_ctx(self.volumes[0])
            idx = torch.empty((H, W), dtype=torch.int64, device=dev)
            fy = torch.empty((H, W), dtype=torch.float32, device=dev)
            fx = torch.empty_like(fy)
            ctx.check(l.dfe_cascade_flow_f32(ctx.handle, _ptr_array(self.probs), rr, n, H, W, maxh, maxw, ptr(idx), None, ptr(fy), ptr(fx)))
        dev = idx.device
        ret = {"index": idx, "confidences": torch.ones((H, W), dtype=torch.float32, device=dev),
               "y": fy.to(torch.int64), "x": fx.to(torch.int64)}
        if process_full:
            hImg, wImg = _g(g, "hImg"), _g(g, "wImg")
            ho, wo = (hImg - H) // 2, (wImg - W) // 2
            if ho == 0 and wo == 0:
                ret["full"] = torch.stack([fy, fx])
                ret["full_confidences"] = ret["confidences"]
            else:
                full = torch.zeros((2, hImg, wImg), dtype=torch.float32, device=dev)
                full[0, ho : ho + H, wo : wo + W] = fy
                full[1, ho : ho + H, wo : wo + W] = fx
                ret["full"] = full
                fc = torch.zeros((hImg, wImg), dtype=torch.float32, device=dev)
                fc[ho : ho + H, wo : wo + W] = 1.0
                ret["full_confidences"] = fc
        return ret


def _ring_join(outs, ratios, maxh, maxw):
    """the "middle remover" + JoinTable(2) on per-pixel windows [P][maxh][maxw] (opticalflow_model_multiscale.lua:293-324):
    scale 1 whole, then per coarser scale the ring blocks top, left, right, bottom."""
    P = outs[0].shape[0]
    parts = [outs[0].reshape(P, -1)]
    for i in range(1, len(ratios)):
        d = int(math.floor(maxw * (ratios[i] - ratios[i - 1]) / (2.0 * ratios[i]) + 0.5))
        o = outs[i]
        parts += [o[:, :d, :].reshape(P, -1), o[:, d : maxh - d, :d].reshape(P, -1), o[:, d : maxh - d, maxw - d :].reshape(P, -1),
                  o[:, maxh - d :, :].reshape(P, -1)]
    return torch.cat(parts, 1)


def getMultiscalePrefilter(geometry, filter):
    """opticalflow_model_multiscale.lua:134-173."""
    assert _g(geometry, "multiscale")
    return MultiscalePrefilter(geometry, filter)


def getModelMultiscale(geometry, full_image=True, prefiltered=False, device="cuda", generator=None, filters="auto"):
    """opticalflow_model_multiscale.lua:175-372.  geometry.layers present (and not prefiltered): getFilter(geometry) in front of
    every scale's matcher -- one stack whose parameters all scales share when geometry.share_filters, independent copies
    (initialised equal, like :clone()) otherwise (:219-226).  No geometry.layers (or filters=None): the raw-patch identity
    filter with geometry.hKernel x wKernel patches."""
    import copy

    from .network import getFilter, _SharedFilter, filter_weights

    assert _g(geometry, "output_extraction_method", "max") == "max"   # :176
    if prefiltered:
        return MultiscaleModel(geometry, None, True)
    layers = _g(geometry, "layers")
    if filters is None or not layers:
        return MultiscaleModel(geometry)
    # the stack's receptive field is what the geometry calls the kernel (opticalflow.lua:154-171)
    kh = 1 + sum(l[2] - 1 for l in layers)
    kw = 1 + sum(l[1] - 1 for l in layers)
    if _g(geometry, "hKernel") is None:
        geometry["hKernel"], geometry["wKernel"] = kh, kw
    assert (_g(geometry, "hKernel"), _g(geometry, "wKernel")) == (kh, kw), "geometry.hKernel / wKernel do not match geometry.layers"
    base = getFilter(geometry, device=device, generator=generator)
    fl = [base]
    for _ in range(1, len(_g(geometry, "ratios"))):
        if _g(geometry, "share_filters", False):
            fl.append(_SharedFilter(base))
        else:
            f = copy.deepcopy(base)
            f.getWeights = (lambda f=f: filter_weights(f))
            fl.append(f)
    return MultiscaleModel(geometry, fl, False)
