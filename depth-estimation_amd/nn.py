"""Stand-ins for the nnx modules on the hot path, with Torch7's nn.Module protocol
(updateOutput / forward, self.output)."""
import torch

from ._lib import lib
from .context import get_ctx, ptr


class Module:
    """Torch7's nn.Module protocol: forward = updateOutput; backward(input, gradOutput, scale) = updateGradInput +
    accGradParameters; parameters() -> (weights, gradWeights); zeroGradParameters; updateParameters(lr)."""

    def __init__(self):
        self.output = None
        self.gradInput = None

    def forward(self, input):
        return self.updateOutput(input)

    def __call__(self, input):
        return self.forward(input)

    def updateGradInput(self, input, gradOutput):
        raise NotImplementedError("%s: updateGradInput" % type(self).__name__)

    def accGradParameters(self, input, gradOutput, scale=1.0):
        pass

    def backward(self, input, gradOutput, scale=1.0):
        gi = self.updateGradInput(input, gradOutput)
        self.accGradParameters(input, gradOutput, scale)
        return gi

    def parameters(self):
        """([weights...], [gradWeights...]) of this module and its children (shared tensors listed once)."""
        ws, gs, seen = [], [], set()
        for m in self._all_modules():
            for w, g in zip(getattr(m, "_params", lambda: [])(), getattr(m, "_grads", lambda: [])()):
                if id(w) not in seen:
                    seen.add(id(w))
                    ws.append(w)
                    gs.append(g)
        return ws, gs

    def _all_modules(self):
        out = [self]
        for m in getattr(self, "modules", []) or []:
            out.extend(m._all_modules() if isinstance(m, Module) else [])
        return out

    def zeroGradParameters(self):
        for g in self.parameters()[1]:
            g.zero_()

    def updateParameters(self, learningRate):
        ws, gs = self.parameters()
        for w, g in zip(ws, gs):
            w.add_(g, alpha=-float(learningRate))


def _f32c(t, what):
    if t.dtype != torch.float32:
        raise TypeError("%s must be a FloatTensor (torch.float32), got %s" % (what, t.dtype))
    return t.contiguous()


class SpatialMatching(Module):
    """nn.SpatialMatching(maxh, maxw, full_output=false) -- un-vendored nnx module.
    Reference call sites: opticalflow_model.lua:93, opticalflow_model_multiscale.lua:216,
    radial/radial_opticalflow_groundtruth.lua:83, tests/time_matching.lua:18.
    forward({in1 KxH1xW1, in2 Kx(H1+maxh-1)x(W1+maxw-1)}) -> H1 x W1 x maxh x maxw."""

    def __init__(self, maxh, maxw, full_output=False):
        super().__init__()
        if full_output:
            raise NotImplementedError("SpatialMatching: the reference only uses full_output=false")
        self.maxh, self.maxw = int(maxh), int(maxw)

    def updateOutput(self, input):
        in1, in2 = input
        # patch 1 is usually prepareInput's narrow of a feature map (opticalflow_model.lua:147-149), i.e. a VIEW: rows and planes at the
        # parent's strides.  It is handed to the matcher as it is (dfe_spatial_matching_strided_f32) instead of being copied first.
        view = (in1.dim() == 3 and in1.dtype == torch.float32 and not in1.is_contiguous() and in1.stride(2) == 1 and in1.stride(1) >= in1.shape[2]
                and in1.stride(0) >= (in1.shape[1] - 1) * in1.stride(1) + in1.shape[2])
        if not view:
            in1 = _f32c(in1, "input[1]")
        in2 = _f32c(in2, "input[2]")
        K, H1, W1 = in1.shape
        if tuple(in2.shape) != (K, H1 + self.maxh - 1, W1 + self.maxw - 1):
            raise ValueError(
                "SpatialMatching: input[2] must be %s for input[1] %s and window %dx%d, got %s"
                % ((K, H1 + self.maxh - 1, W1 + self.maxw - 1), tuple(in1.shape), self.maxh, self.maxw, tuple(in2.shape))
            )
        ctx = get_ctx(in1)
        out = torch.empty((H1, W1, self.maxh, self.maxw), dtype=torch.float32, device=in1.device)
        if view:
            ctx.check(lib().dfe_spatial_matching_strided_f32(ctx.handle, in1.data_ptr(), in1.stride(1), in1.stride(0), ptr(in2), K, H1, W1, self.maxh, self.maxw, ptr(out)))
        else:
            ctx.check(lib().dfe_spatial_matching_f32(ctx.handle, ptr(in1), ptr(in2), K, H1, W1, self.maxh, self.maxw, ptr(out)))
        self.output = out
        return out

    def updateGradInput(self, input, gradOutput):
        """Gradient w.r.t. both feature maps (nnx's updateGradInput; reached from the training drivers through
        model:backward).  Returns [gradIn1, gradIn2]."""
        in1, in2 = input
        in1, in2 = _f32c(in1, "input[1]"), _f32c(in2, "input[2]")
        go = _f32c(gradOutput, "gradOutput")
        K, H1, W1 = in1.shape
        if tuple(go.shape) != (H1, W1, self.maxh, self.maxw):
            raise ValueError("SpatialMatching: gradOutput must be %s, got %s" % ((H1, W1, self.maxh, self.maxw), tuple(go.shape)))
        ctx = get_ctx(in1)
        g1, g2 = torch.empty_like(in1), torch.empty_like(in2)
        ctx.check(lib().dfe_spatial_matching_backward_f32(ctx.handle, ptr(in1), ptr(in2), ptr(go), K, H1, W1, self.maxh, self.maxw, ptr(g1), ptr(g2)))
        self.gradInput = [g1, g2]
        return self.gradInput

    def backward(self, input, gradOutput, scale=1.0):
        return self.updateGradInput(input, gradOutput)


class SpatialRadialMatching(Module):
    """nn.SpatialRadialMatching(hWin) -- un-vendored nnx module.
    Reference call sites: radial/radial_opticalflow_network.lua:33, radial/radial_opticalflow_groundtruth.lua:152.
    forward({in1 KxH1xW, in2 Kx(H1+hWin-1)xW}) -> H1 x W x hWin."""

    def __init__(self, hWin):
        super().__init__()
        self.hWin = int(hWin)

    def updateOutput(self, input):
        in1, in2 = input
        in1, in2 = _f32c(in1, "input[1]"), _f32c(in2, "input[2]")
        K, H1, W = in1.shape
        if tuple(in2.shape) != (K, H1 + self.hWin - 1, W):
            raise ValueError("SpatialRadialMatching: input[2] must be %s, got %s" % ((K, H1 + self.hWin - 1, W), tuple(in2.shape)))
        ctx = get_ctx(in1)
        out = torch.empty((H1, W, self.hWin), dtype=torch.float32, device=in1.device)
        ctx.check(lib().dfe_radial_matching_f32(ctx.handle, ptr(in1), ptr(in2), K, H1, W, self.hWin, ptr(out)))
        self.output = out
        return out

    def updateGradInput(self, input, gradOutput):
        """Gradient w.r.t. both feature maps (radial/train_radial_opticalflow.lua:228-252 trains through it)."""
        in1, in2 = input
        in1, in2 = _f32c(in1, "input[1]"), _f32c(in2, "input[2]")
        go = _f32c(gradOutput, "gradOutput")
        K, H1, W = in1.shape
        if tuple(go.shape) != (H1, W, self.hWin):
            raise ValueError("SpatialRadialMatching: gradOutput must be %s, got %s" % ((H1, W, self.hWin), tuple(go.shape)))
        ctx = get_ctx(in1)
        g1, g2 = torch.empty_like(in1), torch.empty_like(in2)
        ctx.check(lib().dfe_radial_matching_backward_f32(ctx.handle, ptr(in1), ptr(in2), ptr(go), K, H1, W, self.hWin, ptr(g1), ptr(g2)))
        self.gradInput = [g1, g2]
        return self.gradInput

    def backward(self, input, gradOutput, scale=1.0):
        return self.updateGradInput(input, gradOutput)


class SSDCostVolume(Module):
    """unfold(kh,kw) + crop + SpatialMatching(hWin,wWin) on raw frames in one op (A0+A1); the
    reference composes it from three modules, radial/radial_opticalflow_groundtruth.lua:79-84.
    forward({img1 CxHxW, img2 CxHxW}) -> Ho x Wo x hWin x wWin."""

    def __init__(self, hWin, wWin, hKer, wKer):
        super().__init__()
        self.hWin, self.wWin, self.hKer, self.wKer = int(hWin), int(wWin), int(hKer), int(wKer)

    def updateOutput(self, input):
        i0, i1 = input
        i0, i1 = _f32c(i0, "input[1]"), _f32c(i1, "input[2]")
        if i0.shape != i1.shape or i0.dim() != 3:
            raise ValueError("SSDCostVolume: frames must both be CxHxW, got %s and %s" % (tuple(i0.shape), tuple(i1.shape)))
        Cc, H, W = i0.shape
        Ho = H - self.hKer + 1 - self.hWin + 1
        Wo = W - self.wKer + 1 - self.wWin + 1
        ctx = get_ctx(i0)
        out = torch.empty((max(Ho, 0), max(Wo, 0), self.hWin, self.wWin), dtype=torch.float32, device=i0.device)
        ctx.check(
            lib().dfe_ssd_cost_volume_f32(ctx.handle, ptr(i0), ptr(i1), Cc, H, W, self.hKer, self.wKer, self.hWin, self.wWin, ptr(out))
        )
        self.output = out
        return out


def __getattr__(name):  # nn.CascadingAddTable lives in multiscale.py (it needs the codec helpers)
    if name == "CascadingAddTable":
        from .multiscale import CascadingAddTable

        return CascadingAddTable
    if name in ("SmartReshape", "FunctionWrapper", "Mul2", "Log2", "OutputExtractor"):
        from . import glue

        return getattr(glue, name)
    if name in ("SpatialConvolution", "SpatialConvolutionMap", "Tanh", "Sequential", "ParallelTable", "Minus"):
        from . import network

        return getattr(network, name)
    raise AttributeError(name)
