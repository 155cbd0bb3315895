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


class MultiscaleModel(Module):
    """What getModelMultiscale(geometry, full_image=true, prefiltered=false):forward({I0, I1}) computes in
    inference mode for the identity patch filter (opticalflow_model_multiscale.lua:175-333): returns the
    hImg x wImg x nclasses tensor that processOutput consumes.  Also keeps the per-scale cost volumes
    (`self.volumes`, native scale) and probabilities (`self.probs`)."""

    def __init__(self, geometry):
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

    def updateOutput(self, input):
        g = self.geometry
        i0, i1 = input
        i0, i1 = _f32c(i0, "MultiscaleModel: input[1]"), _f32c(i1, "MultiscaleModel: input[2]")   # the C ABI is typed: float32 frames
        Cc, H, W = i0.shape
        maxh, maxw, kh, kw = _g(g, "maxh"), _g(g, "maxw"), _g(g, "hKernel"), _g(g, "wKernel")
        rmax = self.ratios[-1]
        if H % rmax or W % rmax:
            # opticalflow_model_multiscale.lua:234-248: nn.SpatialPadding(0,0,0,0) with pad_b / pad_r up to the next
            # multiple of the coarsest ratio, zeros; the output keeps the padded size (nothing crops it back)
            th, tw = rmax * math.ceil(H / rmax), rmax * math.ceil(W / rmax)
            p0 = torch.zeros((Cc, th, tw), dtype=i0.dtype, device=i0.device)
            p1 = torch.zeros_like(p0)
            p0[:, :H, :W] = i0
            p1[:, :H, :W] = i1
            i0, i1, H, W = p0, p1, th, tw
        ctx = get_ctx(i0)
        l = lib()
        N = maxh * maxw
        self.volumes, self.probs = [], []
        for r in self.ratios:
            vol = torch.empty((H // r, W // r, maxh, maxw), dtype=torch.float32, device=i0.device)
            ctx.check(l.dfe_pyramid_scale_volume_f32(ctx.handle, ptr(i0), ptr(i1), Cc, H, W, r, kh, kw, maxh, maxw, ptr(vol)))
            prob = torch.empty_like(vol)
            ctx.check(l.dfe_softmin_f32(ctx.handle, ptr(vol), vol.numel() // N, N, ptr(prob)))
            self.volumes.append(vol)
            self.probs.append(prob)
        rr, n = ratios_array(self.ratios)
        ncls = l.dfe_multi_nclasses(maxh, maxw, rr, n)
        out = torch.empty((H, W, ncls), dtype=torch.float32, device=i0.device)
        ctx.check(l.dfe_cascade_ring_f32(ctx.handle, _ptr_array(self.probs), rr, n, H, W, maxh, maxw, ptr(out)))
        self.output = out
        return out


    def forwardFlow(self, input, process_full=True, one_call=True, f16_scale=None):
        """model:forward(input) followed by processOutput(geometry, output, process_full) for the 'max' extraction
        without a threshold (opticalflow_model.lua:201-252), fused: the H x W x nclasses tensor is never built
        (dfe_cascade_flow_f32).  Returns the same table: index, confidences (all 1), y, x [, full, full_confidences].
        f16_scale: the per-scale cost volumes are stored as half(cost * f16_scale) (dfe_multiscale_flow_pair_f16; the staged
        path rounds its fp32 volumes to half precision the same way)."""
        g = self.geometry
        i0, i1 = input
        i0, i1 = _f32c(i0, "MultiscaleModel: input[1]"), _f32c(i1, "MultiscaleModel: input[2]")   # the C ABI is typed: float32 frames
        Cc, H, W = i0.shape
        maxh, maxw, kh, kw = _g(g, "maxh"), _g(g, "maxw"), _g(g, "hKernel"), _g(g, "wKernel")
        rmax = self.ratios[-1]
        if H % rmax or W % rmax:   # opticalflow_model_multiscale.lua:234-248
            th, tw = rmax * math.ceil(H / rmax), rmax * math.ceil(W / rmax)
            p0 = torch.zeros((Cc, th, tw), dtype=i0.dtype, device=i0.device)
            p1 = torch.zeros_like(p0)
            p0[:, :H, :W] = i0
            p1[:, :H, :W] = i1
            i0, i1, H, W = p0, p1, th, tw
        ctx = get_ctx(i0)
        l = lib()
        N = maxh * maxw
        rr, n = ratios_array(self.ratios)
        idx = torch.empty((H, W), dtype=torch.int64, device=i0.device)
        if one_call and kh == kw:
            # everything in one C call (dfe_multiscale_flow_pair_f32): no per-scale tensors on the host side
            flow = torch.empty((2, H, W), dtype=torch.float32, device=i0.device)
            if f16_scale:
                ctx.check(l.dfe_multiscale_flow_pair_f16(ctx.handle, ptr(i0), ptr(i1), Cc, H, W, kh, maxh, maxw, rr, n, float(f16_scale), ptr(flow), ptr(idx)))
            else:
                ctx.check(l.dfe_multiscale_flow_pair_f32(ctx.handle, ptr(i0), ptr(i1), Cc, H, W, kh, maxh, maxw, rr, n, ptr(flow), ptr(idx)))
            fy, fx = flow[0], flow[1]
            self.volumes, self.probs = None, None
        else:
            self.volumes, self.probs = [], []
            for r in self.ratios:
                vol = torch.empty((H // r, W // r, maxh, maxw), dtype=torch.float32, device=i0.device)
                ctx.check(l.dfe_pyramid_scale_volume_f32(ctx.handle, ptr(i0), ptr(i1), Cc, H, W, r, kh, kw, maxh, maxw, ptr(vol)))
                if f16_scale:   # what the fp16 volume holds: half(cost * scale) read back as float(stored) * (1 / scale)
                    vol = (vol * float(f16_scale)).to(torch.float16).to(torch.float32) * (1.0 / float(f16_scale))
                prob = torch.empty_like(vol)
                ctx.check(l.dfe_softmin_f32(ctx.handle, ptr(vol), vol.numel() // N, N, ptr(prob)))
                self.volumes.append(vol)
                self.probs.append(prob)
            fy = torch.empty((H, W), dtype=torch.float32, device=i0.device)
            fx = torch.empty_like(fy)
            ctx.check(l.dfe_cascade_flow_f32(ctx.handle, _ptr_array(self.probs), rr, n, H, W, maxh, maxw, ptr(idx), None, ptr(fy), ptr(fx)))
        ret = {"index": idx, "confidences": torch.ones((H, W), dtype=torch.float32, device=i0.device),
               "y": fy.to(torch.int64), "x": fx.to(torch.int64)}
        if process_full:
            hImg, wImg = _g(g, "hImg"), _g(g, "wImg")
            ho, wo = (hImg - H) // 2, (wImg - W) // 2
            if ho == 0 and wo == 0:
                ret["full"] = torch.stack([fy, fx])
                ret["full_confidences"] = ret["confidences"]
            else:
                full = torch.zeros((2, hImg, wImg), dtype=torch.float32, device=i0.device)
                full[0, ho : ho + H, wo : wo + W] = fy
                full[1, ho : ho + H, wo : wo + W] = fx
                ret["full"] = full
                fc = torch.zeros((hImg, wImg), dtype=torch.float32, device=i0.device)
                fc[ho : ho + H, wo : wo + W] = 1.0
                ret["full_confidences"] = fc
        return ret


def getModelMultiscale(geometry, full_image=True, prefiltered=False):
    """opticalflow_model_multiscale.lua:175 (inference mode, identity patch filter)."""
    if prefiltered:
        raise NotImplementedError("prefiltered (learned filter) inputs are next-row N1")
    return MultiscaleModel(geometry)
