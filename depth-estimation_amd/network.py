"""The learned patch-feature stack and the single-scale model of the reference (next-row N1):
getFilter / getModel of opticalflow_model.lua:45-130 and getFilter of radial/radial_opticalflow_network.lua:6-30, with
nn.SpatialConvolution / nn.SpatialConvolutionMap / nn.Tanh / nn.Sequential / nn.ParallelTable stand-ins that keep Torch7's
module protocol (`modules` is a plain index-assignable list, as the callers patch it by index)."""
import math

import torch

from ._lib import lib
from .context import get_ctx, ptr
from .nn import Module, SpatialMatching, _f32c
from . import glue


class SpatialConvolution(Module):
    """nn.SpatialConvolution(nInputPlane, nOutputPlane, kW, kH): valid cross-correlation + bias."""

    def __init__(self, nInputPlane, nOutputPlane, kW, kH, device="cuda", generator=None):
        super().__init__()
        self.nInputPlane, self.nOutputPlane, self.kW, self.kH = int(nInputPlane), int(nOutputPlane), int(kW), int(kH)
        stdv = 1.0 / math.sqrt(self.kW * self.kH * self.nInputPlane)   # nn.SpatialConvolution:reset
        self.weight = (torch.rand((self.nOutputPlane, self.nInputPlane, self.kH, self.kW), generator=generator) * 2 - 1).mul_(stdv).to(device)
        self.bias = (torch.rand((self.nOutputPlane,), generator=generator) * 2 - 1).mul_(stdv).to(device)
        self.gradWeight, self.gradBias = torch.zeros_like(self.weight), torch.zeros_like(self.bias)

    def _params(self):
        return [self.weight, self.bias]

    def _grads(self):
        return [self.gradWeight, self.gradBias]

    def updateGradInput(self, input, gradOutput):
        x, go = _f32c(input, "input"), _f32c(gradOutput, "gradOutput")
        nIn, H, W = x.shape
        if tuple(go.shape) != (self.nOutputPlane, H - self.kH + 1, W - self.kW + 1):
            raise ValueError("SpatialConvolution: gradOutput must be %s, got %s" % ((self.nOutputPlane, H - self.kH + 1, W - self.kW + 1), tuple(go.shape)))
        gi = torch.empty_like(x)
        ctx = get_ctx(x)
        ctx.check(lib().dfe_spatial_convolution_grad_input_f32(ctx.handle, ptr(go), ptr(self.weight), nIn, self.nOutputPlane, H, W, self.kH, self.kW, ptr(gi)))
        self.gradInput = gi
        return gi

    def accGradParameters(self, input, gradOutput, scale=1.0):
        x, go = _f32c(input, "input"), _f32c(gradOutput, "gradOutput")
        nIn, H, W = x.shape
        ctx = get_ctx(x)
        ctx.check(lib().dfe_spatial_convolution_acc_grad_f32(ctx.handle, ptr(x), ptr(go), nIn, self.nOutputPlane, H, W, self.kH, self.kW, float(scale),
                                                             ptr(self.gradWeight), ptr(self.gradBias)))

    # kernel = "exact": the direct kernel, bit-identical to the CPU loop (separately rounded multiply and add);
    # kernel = "mfma": implicit GEMM on the matrix cores (fused multiply-adds in the same order: <= 1e-5 relative)
    kernel = "exact"

    def updateOutput(self, input):
        x = _f32c(input, "input")
        nIn, H, W = x.shape
        if nIn != self.nInputPlane:
            raise ValueError("SpatialConvolution: expected %d input planes, got %d" % (self.nInputPlane, nIn))
        out = torch.empty((self.nOutputPlane, H - self.kH + 1, W - self.kW + 1), dtype=torch.float32, device=x.device)
        ctx = get_ctx(x)
        if self.kernel == "mfma":
            ctx.check(lib().dfe_spatial_convolution_mfma_f32(ctx.handle, ptr(x), ptr(self.weight), ptr(self.bias), nIn, self.nOutputPlane, H, W,
                                                             self.kH, self.kW, 0, ptr(out)))
            self.output = out
            return out
        ctx.check(lib().dfe_spatial_convolution_f32(ctx.handle, ptr(x), ptr(self.weight), ptr(self.bias), nIn, self.nOutputPlane, H, W,
                                                    self.kH, self.kW, ptr(out)))
        self.output = out
        return out


def tables_random(nin, nout, nto, generator=None):
    """nn.tables.random(nin, nout, nto): every output plane is connected to `nto` distinct random input planes; rows are
    (from, to), 1-based."""
    rows = []
    for o in range(1, nout + 1):
        for i in torch.randperm(nin, generator=generator)[:nto].tolist():
            rows.append((i + 1, o))
    return torch.tensor(rows, dtype=torch.int32)


class SpatialConvolutionMap(Module):
    """nn.SpatialConvolutionMap(connTable, kW, kH)."""

    def __init__(self, connTable, kW, kH, device="cuda", generator=None):
        super().__init__()
        self.connTable = connTable.to(torch.int32).contiguous()
        self.kW, self.kH = int(kW), int(kH)
        self.nInputPlane = int(self.connTable[:, 0].max())
        self.nOutputPlane = int(self.connTable[:, 1].max())
        nconn = self.connTable.shape[0]
        ninp = nconn / self.nOutputPlane
        stdv = 1.0 / math.sqrt(self.kW * self.kH * ninp)
        self.weight = (torch.rand((nconn, self.kH, self.kW), generator=generator) * 2 - 1).mul_(stdv).to(device)
        self.bias = (torch.rand((self.nOutputPlane,), generator=generator) * 2 - 1).mul_(stdv).to(device)
        self._conn_dev = self.connTable.to(device)
        self.gradWeight, self.gradBias = torch.zeros_like(self.weight), torch.zeros_like(self.bias)

    def _params(self):
        return [self.weight, self.bias]

    def _grads(self):
        return [self.gradWeight, self.gradBias]

    def updateGradInput(self, input, gradOutput):
        x, go = _f32c(input, "input"), _f32c(gradOutput, "gradOutput")
        nIn, H, W = x.shape
        if tuple(go.shape) != (self.nOutputPlane, H - self.kH + 1, W - self.kW + 1):
            raise ValueError("SpatialConvolutionMap: gradOutput must be %s, got %s" % ((self.nOutputPlane, H - self.kH + 1, W - self.kW + 1), tuple(go.shape)))
        gi = torch.empty_like(x)
        ctx = get_ctx(x)
        ctx.check(lib().dfe_spatial_convolution_map_grad_input_f32(ctx.handle, ptr(go), ptr(self.weight), ptr(self._conn_dev), self.connTable.shape[0], nIn,
                                                                   self.nOutputPlane, H, W, self.kH, self.kW, ptr(gi)))
        self.gradInput = gi
        return gi

    def accGradParameters(self, input, gradOutput, scale=1.0):
        x, go = _f32c(input, "input"), _f32c(gradOutput, "gradOutput")
        nIn, H, W = x.shape
        ctx = get_ctx(x)
        ctx.check(lib().dfe_spatial_convolution_map_acc_grad_f32(ctx.handle, ptr(x), ptr(go), ptr(self._conn_dev), self.connTable.shape[0], nIn,
                                                                 self.nOutputPlane, H, W, self.kH, self.kW, float(scale), ptr(self.gradWeight), ptr(self.gradBias)))

    def updateOutput(self, input):
        x = _f32c(input, "input")
        nIn, H, W = x.shape
        if nIn < self.nInputPlane:
            raise ValueError("SpatialConvolutionMap: the table reads plane %d, input has %d" % (self.nInputPlane, nIn))
        out = torch.empty((self.nOutputPlane, H - self.kH + 1, W - self.kW + 1), dtype=torch.float32, device=x.device)
        ctx = get_ctx(x)
        ctx.check(lib().dfe_spatial_convolution_map_f32(ctx.handle, ptr(x), ptr(self.weight), ptr(self.bias), ptr(self._conn_dev),
                                                        self.connTable.shape[0], nIn, self.nOutputPlane, H, W, self.kH, self.kW, ptr(out)))
        self.output = out
        return out


class Tanh(Module):
    def updateOutput(self, input):
        x = _f32c(input, "input")
        out = torch.empty_like(x)
        ctx = get_ctx(x)
        ctx.check(lib().dfe_tanh_f32(ctx.handle, ptr(x), x.numel(), ptr(out)))
        self.output = out
        return out

    def updateGradInput(self, input, gradOutput):
        go = _f32c(gradOutput, "gradOutput")
        gi = torch.empty_like(go)
        ctx = get_ctx(go)
        ctx.check(lib().dfe_tanh_backward_f32(ctx.handle, ptr(self.output), ptr(go), go.numel(), ptr(gi)))
        self.gradInput = gi
        return gi


def gaussian1D(size, sigma=0.25, amplitude=1.0, normalize=False):
    """image.gaussian1D (un-vendored `image`; as recalled): amplitude * exp(-((i - center) / (sigma * size))^2 / 2), 1-based i,
    center = size / 2 + 0.5."""
    i = torch.arange(1, size + 1, dtype=torch.float64)
    g = amplitude * torch.exp(-(((i - (size / 2 + 0.5)) / (sigma * size)) ** 2) / 2)
    if normalize:
        g = g / g.sum()
    return g.to(torch.float32)


class SpatialContrastiveNormalization(Module):
    """nn.SpatialContrastiveNormalization(nInputPlane, kernel, threshold, thresval) with a 1-D kernel -- the front end of both
    filter branches in version2/network.lua:12,22."""

    def __init__(self, nInputPlane=1, kernel=None, threshold=1e-4, thresval=1e-4):
        super().__init__()
        self.nInputPlane = int(nInputPlane)
        self.kernel = (torch.ones(9) if kernel is None else kernel).to(torch.float32).cpu().contiguous()
        if self.kernel.dim() != 1:
            raise NotImplementedError("SpatialContrastiveNormalization: the reference uses a 1-D kernel (image.gaussian1D)")
        self.threshold, self.thresval = float(threshold), float(thresval)

    def updateOutput(self, input):
        x = _f32c(input, "input")
        Cc, H, W = x.shape
        if Cc != self.nInputPlane:
            raise ValueError("SpatialContrastiveNormalization: expected %d planes, got %d" % (self.nInputPlane, Cc))
        out = torch.empty_like(x)
        ctx = get_ctx(x)
        import ctypes
        kp = self.kernel.numpy().ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        ctx.check(lib().dfe_contrastive_normalization_f32(ctx.handle, ptr(x), Cc, H, W, kp, self.kernel.numel(), self.threshold, self.thresval, ptr(out)))
        self.output = out
        return out


class Sequential(Module):
    def __init__(self):
        super().__init__()
        self.modules = []

    def add(self, m):
        self.modules.append(m)
        return self

    # nn.SpatialConvolution directly followed by nn.Tanh (every layer of getFilter but the last) runs as ONE launch -- the convolution
    # kernel's epilogue applies the tanh, same bits as the two modules -- so the pair's pre-activation is not materialised: the
    # convolution module's .output stays None (nothing on the path reads it: Tanh's gradient needs its own output only).  fuse = False on
    # the instance gives the module-by-module evaluation back.
    fuse = True

    def updateOutput(self, input):
        self._inputs = []
        mods = self.modules
        i = 0
        while i < len(mods):
            m = mods[i]
            self._inputs.append(input)
            if (self.fuse and i + 1 < len(mods) and isinstance(mods[i + 1], Tanh) and isinstance(m, SpatialConvolution)
                    and not isinstance(m, SpatialConvolutionMap) and m.kernel == "exact" and input.dim() == 3 and input.shape[0] == m.nInputPlane):
                x = _f32c(input, "input")
                nIn, H, W = x.shape
                out = torch.empty((m.nOutputPlane, H - m.kH + 1, W - m.kW + 1), dtype=torch.float32, device=x.device)
                ctx = get_ctx(x)
                ctx.check(lib().dfe_spatial_convolution_tanh_f32(ctx.handle, ptr(x), ptr(m.weight), ptr(m.bias), nIn, m.nOutputPlane, H, W, m.kH, m.kW, ptr(out)))
                m.output = None
                mods[i + 1].output = out
                self._inputs.append(None)
                input = out
                i += 2
                continue
            input = m.forward(input)
            i += 1
        self.output = input
        return input

    def backward(self, input, gradOutput, scale=1.0):
        """nn.Sequential:backward: modules in reverse order, each on the input it saw in the last forward."""
        g = gradOutput
        for m, x in zip(reversed(self.modules), reversed(self._inputs)):
            g = m.backward(x, g, scale)
        self.gradInput = g
        return g

    def updateGradInput(self, input, gradOutput):
        g = gradOutput
        for m, x in zip(reversed(self.modules), reversed(self._inputs)):
            g = m.updateGradInput(x, g)
        self.gradInput = g
        return g


class ParallelTable(Module):
    def __init__(self):
        super().__init__()
        self.modules = []

    def add(self, m):
        self.modules.append(m)
        return self

    def updateOutput(self, input):
        self.output = [m.forward(x) for m, x in zip(self.modules, input)]
        return self.output

    def backward(self, input, gradOutput, scale=1.0):
        self.gradInput = [m.backward(x, g, scale) for m, x, g in zip(self.modules, input, gradOutput)]
        return self.gradInput

    def updateGradInput(self, input, gradOutput):
        self.gradInput = [m.updateGradInput(x, g) for m, x, g in zip(self.modules, input, gradOutput)]
        return self.gradInput


class Minus(Module):
    def updateOutput(self, input):
        self.output = -input
        return self.output

    def updateGradInput(self, input, gradOutput):
        self.gradInput = -gradOutput
        return self.gradInput


class SoftMaxWindow(Module):
    """The FunctionWrapper of getModel (opticalflow_model.lua:96-109) applied after nn.Minus: SmartReshape({-1,-2},{-3,-4})
    -> nn.SoftMax -> SmartReshape(H, W, -2).  Together with the Minus in front this is A3's softmin; here the pair
    (Minus, SoftMaxWindow) is evaluated by dfe_softmin_f32 on the un-negated costs."""

    def updateOutput(self, input):
        x = _f32c(input, "input")
        H, W = x.shape[0], x.shape[1]
        N = x.numel() // (H * W)
        cost = (-x).contiguous()          # undo the Minus that precedes this module
        out = torch.empty((H, W, N), dtype=torch.float32, device=x.device)
        ctx = get_ctx(x)
        ctx.check(lib().dfe_softmin_f32(ctx.handle, ptr(cost), H * W, N, ptr(out)))
        self.output = out
        return out

    def updateGradInput(self, input, gradOutput):
        go = _f32c(gradOutput, "gradOutput")
        out = self.output
        gi = torch.empty_like(out)
        ctx = get_ctx(out)
        N = out.shape[-1]
        ctx.check(lib().dfe_softmax_backward_f32(ctx.handle, ptr(out), ptr(go.reshape(out.shape)), out.numel() // N, N, ptr(gi)))
        self.gradInput = gi.reshape(input.shape)
        return self.gradInput


def getFilter(geometry, device="cuda", generator=None):
    """opticalflow_model.lua:45-79: geometry.layers[i] = {nIn, kW, kH, nOut}; Tanh between layers; a layer whose fan-in
    differs from the previous fan-out becomes a SpatialConvolutionMap over a random connection table."""
    layers = geometry["layers"] if isinstance(geometry, dict) else geometry.layers
    filt = Sequential()
    for i, l in enumerate(layers):
        nin, kw, kh, nout = l
        if i == 0 or layers[i - 1][3] == nin:
            filt.add(SpatialConvolution(nin, nout, kw, kh, device=device, generator=generator))
        else:
            filt.add(SpatialConvolutionMap(tables_random(layers[i - 1][3], nout, nin, generator=generator), kw, kh, device=device, generator=generator))
        if i != len(layers) - 1:
            filt.add(Tanh())
    filt.getWeights = lambda: filter_weights(filt)
    return filt


def getFilterRadial(networkp, device="cuda", generator=None):
    """radial/radial_opticalflow_network.lua:6-30: layers are 'tanh' or {nIn, kH, kW, nOut} (note the kH, kW order)."""
    filt = Sequential()
    last = None
    for layer in networkp["layers"]:
        if layer == "tanh":
            filt.add(Tanh())
        elif isinstance(layer, (list, tuple)):
            nin, kh, kw, nout = layer
            if last is None or nin == last:
                filt.add(SpatialConvolution(nin, nout, kw, kh, device=device, generator=generator))
            else:
                filt.add(SpatialConvolutionMap(tables_random(last, nout, nin, generator=generator), kw, kh, device=device, generator=generator))
            last = nout
        else:
            raise ValueError("Unknown layer %r" % (layer,))
    return filt


_SHARED_ATTRS = ("weight", "bias", "gradWeight", "gradBias", "kernel", "connTable", "_conn_dev")


def shared_clone(src):
    """module:clone('weight', 'bias', 'gradWeight', 'gradBias') of ONE convolution: a second module of the same class whose
    parameters, gradient buffers and kernel choice ARE the source's -- read through the source at call time, so rebinding
    `src.weight = ...` (how trained weights get loaded), `src.kernel = "mfma"` or a device move shows in the clone, and
    assignments to the clone's parameters land in the source -- while outputs / gradInputs stay its own."""
    cls = type(src)
    ns = {"__getattr__": lambda self, name: getattr(object.__getattribute__(self, "_src"), name)}
    for n in _SHARED_ATTRS:
        ns[n] = property(lambda self, n=n: getattr(object.__getattribute__(self, "_src"), n),
                         lambda self, v, n=n: setattr(object.__getattribute__(self, "_src"), n, v))
    proxy_cls = type("Shared" + cls.__name__, (cls,), ns)
    p = proxy_cls.__new__(proxy_cls)
    object.__setattr__(p, "_src", src)
    p.output = p.gradInput = None
    return p


class _SharedFilter(Module):
    """filter:clone('weight','bias','gradWeight','gradBias') -- the second branch shares the first one's parameters AND
    gradient buffers (accGradParameters of both branches accumulates into the same tensors), but keeps its own outputs."""

    def __init__(self, filt):
        super().__init__()
        self.filt = filt
        self.modules = [shared_clone(m) if isinstance(m, (SpatialConvolution, SpatialConvolutionMap)) else type(m)() for m in filt.modules]

    fuse = True
    updateOutput = Sequential.updateOutput          # (with its convolution + tanh pairs in one launch)

    def backward(self, input, gradOutput, scale=1.0):
        g = gradOutput
        for m, x in zip(reversed(self.modules), reversed(self._inputs)):
            g = m.backward(x, g, scale)
        self.gradInput = g
        return g

    updateGradInput = Sequential.updateGradInput

    def getWeights(self):
        return filter_weights(self)


def filter_weights(filt):
    """filter:getWeights() of getFilter (opticalflow_model.lua:66-76): {'layer<i>': weight} over the modules that have one."""
    out, i = {}, 1
    for m in filt.modules:
        if isinstance(m, (SpatialConvolution, SpatialConvolutionMap)):
            out["layer%d" % i] = m.weight
            i += 1
    return out


def getModel(geometry, full_image=True, prefiltered=False, device="cuda", generator=None):
    """opticalflow_model.lua:81-130 (single scale): [ParallelTable(filter, shared clone)] -> SpatialMatching -> Minus ->
    softmax over the window -> OutputExtractor for 'mean'.  `model.modules` is a plain list in the reference's order, so
    callers can patch it by index (depth_estimation_api.lua:27)."""
    g = geometry
    get = (lambda k, d=None: g.get(k, d)) if isinstance(g, dict) else (lambda k, d=None: getattr(g, k, d))
    model = Sequential()
    if not prefiltered:
        filt = getFilter(g, device=device, generator=generator)
        par = ParallelTable()
        par.add(filt)
        par.add(_SharedFilter(filt))
        model.add(par)
    model.add(SpatialMatching(get("maxh"), get("maxw"), False))
    model.add(Minus())
    model.add(SoftMaxWindow())
    if get("output_extraction_method", "max") == "mean":
        model.add(glue.OutputExtractor(get("maxh"), get("maxw")))
    elif get("training_mode", False):
        model.add(glue.Log2(1e-10))
    # model:getWeights() (:119-125): the first filter branch's, nothing when prefiltered
    model.getWeights = (lambda: {}) if prefiltered else (lambda: filter_weights(model.modules[0].modules[0]))
    model.forwardFlow = lambda input, threshold=None, one_call=True: single_scale_forward_flow(model, g, input, threshold, one_call, prefiltered)
    return model


def single_scale_forward_flow(model, geometry, input, threshold=None, one_call=True, prefiltered=False):
    """What the drivers do with a single-scale model per frame pair (depth_estimation_opticalflow.lua:103-116, depth_estimation_api.lua:
    164-168): input = prepareInput(geometry, patch1, patch2); moutput = model:forward(input); processOutput(geometry, moutput, true,
    threshold).  `input` here is the PAIR BEFORE prepareInput -- frames [C][H][W] for getModel(geometry, true, false), feature maps for
    the prefiltered model -- because the narrow is part of what the one-call entry fuses.
      one_call=True: dfe_flow_pair_filtered_f32 -- filter stack of both frames, the narrow, matcher, Minus / SoftMax, the arg-max with the
        centre tie-break or extractOutput, decode and centre paste; with 16- / 17-wide windows the volume is never written;
      one_call=False: the modules, then processOutput -- the same results bit for bit.
    Returns processOutput's table: index, confidences, y, x, full, full_confidences (+ scores with a threshold)."""
    from .opticalflow_model import prepareInput, processOutput, _g
    from .multiscale import filter_layers_array

    a, b = input
    g = geometry
    maxh, maxw = _g(g, "maxh"), _g(g, "maxw")
    if _g(g, "output_extraction_method", "max") != "max" or len(model.modules) != (3 if prefiltered else 4) or not one_call:
        # ('mean' extraction, training mode, or a model whose module list a caller has patched: module by module)
        if prefiltered:
            inp = prepareInput(dict(g, prefilter=True) if isinstance(g, dict) else g, a, b)
        else:
            # the narrow applies to the FEATURES (prepareInput is called on filter outputs, depth_estimation_opticalflow.lua:66-106): the
            # frames go through the model's own filter branches whole and patch 1's features are narrowed in between
            par = model.modules[0]
            f1, f2 = par.modules[0].forward(a), par.modules[1].forward(b)
            y0, x0 = math.ceil(maxh / 2) - 1, math.ceil(maxw / 2) - 1
            inp = [f1[:, y0 : y0 + f1.shape[1] - maxh + 1, x0 : x0 + f1.shape[2] - maxw + 1], f2]
        out = inp
        for m in (model.modules if prefiltered else model.modules[1:]):
            out = m.forward(out)
        return processOutput(g, out, True, threshold)
    if a.dtype != torch.float32 or b.dtype != torch.float32 or tuple(a.shape) != tuple(b.shape) or a.dim() != 3:
        raise TypeError("forwardFlow: two float32 C x H x W tensors of one size expected")
    a, b = a.contiguous(), b.contiguous()
    Cc, H, W = a.shape
    if prefiltered:
        arr, nl, keep, hk, wk = None, 0, None, 1, 1
    else:
        arr, nl, keep = filter_layers_array([model.modules[0].modules[0]])
        layers = _g(g, "layers")
        hk, wk = 1 + sum(l[2] - 1 for l in layers), 1 + sum(l[1] - 1 for l in layers)
    H1, W1 = H - hk + 1 - maxh + 1, W - wk + 1 - maxw + 1
    if H1 <= 0 or W1 <= 0:
        raise ValueError("forwardFlow: frame %dx%d too small" % (H, W))
    hImg, wImg = _g(g, "hImg"), _g(g, "wImg")
    dev = a.device
    full = torch.empty((2, hImg, wImg), dtype=torch.float32, device=dev)
    fc = torch.empty((hImg, wImg), dtype=torch.float32, device=dev)
    idx = torch.empty((H1, W1), dtype=torch.int64, device=dev)
    sc = torch.empty((H1, W1), dtype=torch.float32, device=dev) if threshold is not None else None
    ctx = get_ctx(a)
    ctx.check(lib().dfe_flow_pair_filtered_f32(ctx.handle, ptr(a), ptr(b), Cc, H, W, arr, nl, maxh, maxw, 0 if threshold is None else 1,
                                               float(threshold or 0.0), hImg, wImg, ptr(full), ptr(fc), ptr(idx), ptr(sc)))
    del keep
    ho, wo = (hImg - H1) // 2, (wImg - W1) // 2
    ret = {"index": idx, "full": full, "full_confidences": fc,
           "y": full[0, ho : ho + H1, wo : wo + W1].to(torch.int64), "x": full[1, ho : ho + H1, wo : wo + W1].to(torch.int64),
           "confidences": fc[ho : ho + H1, wo : wo + W1]}
    if sc is not None:
        ret["scores"] = sc
    return ret
