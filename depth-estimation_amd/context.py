"""dfe_ctx wrapper: one context per (device, stream)."""
import ctypes as C

from ._lib import lib, DfeError, DFE_OK

_ctxs = {}


class Context:
    """Owns a dfe_ctx bound to a HIP device and stream (include/dfe.h: dfe_ctx_create)."""

    def __init__(self, device=0, stream=None, own_stream=False):
        """stream: a hipStream_t as an integer (0/None = the default stream, torch's default);
        own_stream=True makes the ctx create a private non-blocking stream instead."""
        l = lib()
        h = C.c_void_p()
        rc = l.dfe_ctx_create(int(device), C.c_void_p(stream) if stream else None, 1 if own_stream else 0, C.byref(h))
        if rc != DFE_OK:
            raise DfeError(rc, l.dfe_last_error(None).decode())
        self.handle = h
        self.device = int(device)

    def check(self, rc):
        if rc != DFE_OK:
            raise DfeError(rc, lib().dfe_last_error(self.handle).decode())

    def synchronize(self):
        self.check(lib().dfe_ctx_synchronize(self.handle))

    def set_cost_volume_kernel(self, mode):
        """0 auto, 1 reference-order kernel, 2 tiled kernel only."""
        self.check(lib().dfe_set_cost_volume_kernel(self.handle, int(mode)))

    def set_cost_volume_tile(self, tyq):
        """0 = tile height chosen per shape; 2..5 forces it (tuning / tests)."""
        self.check(lib().dfe_set_cost_volume_tile(self.handle, int(tyq)))

    def set_option(self, key, value):
        """dfe_set_option: force (>= 0) or release (-1 / None) one of the launchers' behaviour switches (include/dfe.h)."""
        self.check(lib().dfe_set_option(self.handle, str(key).encode(), -1 if value is None else int(value)))

    def get_option(self, key):
        v = C.c_int()
        self.check(lib().dfe_get_option(self.handle, str(key).encode(), C.byref(v)))
        return v.value

    def options(self, **kw):
        """with ctx.options(fine_fuse=1, mid_fuse=0): ...  -- the switches are restored on exit."""
        ctx = self

        class _Scope:
            def __enter__(self_s):
                self_s.old = {k: ctx.get_option(k) for k in kw}
                for k, v in kw.items():
                    ctx.set_option(k, v)
                return ctx

            def __exit__(self_s, *exc):
                for k, v in self_s.old.items():
                    ctx.set_option(k, v)
                return False

        return _Scope()

    def last_kernel(self):
        return lib().dfe_last_kernel(self.handle).decode()

    def close(self):
        if self.handle:
            lib().dfe_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def get_ctx(tensor_or_device=None):
    """Context on the tensor's device, enqueuing on torch's current stream for that device so
    libdfe kernels are ordered with the caller's torch work."""
    import torch

    if tensor_or_device is None:
        dev = torch.cuda.current_device()
    elif hasattr(tensor_or_device, "device"):
        if tensor_or_device.device.type != "cuda":
            raise DfeError(-1, "tensor is on %s; libdfe takes device tensors only (no CPU fallback)" % tensor_or_device.device)
        dev = tensor_or_device.device.index
    else:
        dev = int(tensor_or_device)
    stream = torch.cuda.current_stream(dev).cuda_stream
    key = (dev, stream)
    if key not in _ctxs:
        _ctxs[key] = Context(dev, stream)
    return _ctxs[key]


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None
