"""Torch7 binary serialisation (torch.save / torch.load of the reference's era), read and written without Torch7.

The data formats either side of the hot path are Torch7 files: the camera calibration tables `radial/*.cal`,
`version2/rectified_gopro.cal` (loaded by `torch.load(calibrationp)`, radial/radial_opticalflow_data.lua:160-170,
depth_estimation_api.lua:33-38) and the trained filter weights written by `saveModel` / `saveNetwork`
(opticalflow_model_io.lua:149-163, radial/radial_opticalflow_network.lua:122-156: a table with `weights`, `geometry`,
`learning`, ... and Lua closures).  Format (little endian, "binary" mode of torch.File):

    object   := int32 tag, payload
    tag 0 nil | 1 number: float64 | 2 string: int32 n, n bytes | 5 boolean: int32
    tag 3 table:  int32 index; first occurrence: int32 n, n x (key object, value object); later occurrences: the index only
    tag 4 torch:  int32 index; first occurrence: string "V 1" (newer files; absent in the oldest), string class name, then
                  torch.XTensor:  int32 ndim, ndim x int64 size, ndim x int64 stride, int64 storage offset (1-based), object storage
                  torch.XStorage: int64 n, n raw elements
    tag 6 / 7 / 8 function (7 = legacy, 8 = current TYPE_RECUR_FUNCTION): int32 index; first occurrence: string dumped chunk, object upvalues (kept opaque)

Tables whose keys are exactly 1..n come back as lists, other tables as dicts; tensors as numpy arrays (a copy with the
file's sizes and strides applied); closures as Torch7Function(bytecode) placeholders.  `save` writes numbers, strings,
booleans, lists / dicts and numpy arrays back in the same format (enough for weights + geometry files a Torch7 reader loads).
"""
import struct

import numpy as np

_STORAGE_DTYPES = {
    "torch.FloatStorage": np.float32, "torch.DoubleStorage": np.float64, "torch.LongStorage": np.int64,
    "torch.IntStorage": np.int32, "torch.ShortStorage": np.int16, "torch.ByteStorage": np.uint8, "torch.CharStorage": np.int8,
}
_TENSOR_STORAGE = {k.replace("Storage", "Tensor"): k for k in _STORAGE_DTYPES}
_DTYPE_TENSOR = {np.dtype(v): k.replace("Storage", "Tensor") for k, v in _STORAGE_DTYPES.items()}


class Torch7Function:
    """A serialised Lua closure (string.dump bytecode + upvalues); opaque here."""

    def __init__(self, bytecode, upvalues=None):
        self.bytecode, self.upvalues = bytecode, upvalues

    def __repr__(self):
        return "Torch7Function(%d bytes)" % len(self.bytecode)


class Torch7Object:
    """A torch class instance this reader has no decoder for (kept as class name + the table it serialised, e.g. nn modules)."""

    def __init__(self, typename, fields):
        self.typename, self.fields = typename, fields

    def __repr__(self):
        return "Torch7Object(%s)" % self.typename


class _Reader:
    def __init__(self, data):
        self.b, self.p, self.memo = data, 0, {}

    def take(self, fmt):
        n = struct.calcsize(fmt)
        if self.p + n > len(self.b):
            raise ValueError("torch7_io: truncated file at byte %d" % self.p)
        v = struct.unpack_from("<" + fmt, self.b, self.p)
        self.p += n
        return v[0] if len(v) == 1 else v

    def string(self):
        n = self.take("i")
        if n < 0 or self.p + n > len(self.b):
            raise ValueError("torch7_io: bad string length %d at byte %d" % (n, self.p))
        s = self.b[self.p : self.p + n]
        self.p += n
        return s

    def obj(self):
        tag = self.take("i")
        if tag == 0:
            return None
        if tag == 1:
            return self.take("d")
        if tag == 2:
            return self.string().decode("latin-1")
        if tag == 5:
            return self.take("i") != 0
        if tag == 3:
            idx = self.take("i")
            if idx in self.memo:
                return self.memo[idx]
            n = self.take("i")
            d = {}
            self.memo[idx] = d
            for _ in range(n):
                k = self.obj()
                d[k] = self.obj()
            keys = list(d.keys())
            if keys and all(isinstance(k, float) and k == int(k) for k in keys) and sorted(int(k) for k in keys) == list(range(1, len(keys) + 1)):
                lst = [d[float(i)] for i in range(1, len(keys) + 1)]
                self.memo[idx] = lst
                return lst
            return d
        if tag == 4:
            idx = self.take("i")
            if idx in self.memo:
                return self.memo[idx]
            name = self.string().decode("latin-1")
            if name.startswith("V "):                     # version marker of newer files, then the class name
                name = self.string().decode("latin-1")
            if name in _STORAGE_DTYPES:
                n = self.take("q")
                dt = np.dtype(_STORAGE_DTYPES[name]).newbyteorder("<")
                if n < 0 or self.p + n * dt.itemsize > len(self.b):
                    raise ValueError("torch7_io: bad storage size %d" % n)
                a = np.frombuffer(self.b, dt, n, self.p).astype(_STORAGE_DTYPES[name])
                self.p += n * dt.itemsize
                self.memo[idx] = a
                return a
            if name in _TENSOR_STORAGE:
                nd = self.take("i")
                if nd < 0 or nd > 64:
                    raise ValueError("torch7_io: tensor with %d dimensions" % nd)
                size = [self.take("q") for _ in range(nd)]
                stride = [self.take("q") for _ in range(nd)]
                off = self.take("q") - 1
                st = self.obj()
                if st is None or nd == 0:
                    t = np.zeros([0] * max(nd, 1), _STORAGE_DTYPES[_TENSOR_STORAGE[name]])
                else:
                    # the file's geometry is untrusted: every size / stride non-negative and the last element inside the storage
                    if not isinstance(st, np.ndarray) or st.dtype != np.dtype(_STORAGE_DTYPES[_TENSOR_STORAGE[name]]):
                        raise ValueError("torch7_io: %s over a %s" % (name, type(st).__name__))
                    if off < 0 or any(x < 0 for x in size) or any(x < 0 for x in stride):
                        raise ValueError("torch7_io: tensor with negative offset / size / stride (%d, %s, %s)" % (off, size, stride))
                    if all(x > 0 for x in size):
                        last = off + sum((n - 1) * x for n, x in zip(size, stride))
                        if last >= len(st):
                            raise ValueError("torch7_io: tensor reaches element %d of a storage of %d" % (last, len(st)))
                        t = np.lib.stride_tricks.as_strided(st[off:], shape=size, strides=[x * st.itemsize for x in stride]).copy()
                    else:
                        t = np.zeros(size, st.dtype)
                self.memo[idx] = t
                return t
            o = Torch7Object(name, None)                  # any other torch class serialises as its table of fields
            self.memo[idx] = o
            o.fields = self.obj()
            return o
        if tag in (6, 7, 8):     # legacy function / recursive function (old and current Torch7 File.lua numbering)
            idx = self.take("i")
            if idx in self.memo:
                return self.memo[idx]
            f = Torch7Function(self.string())
            self.memo[idx] = f
            f.upvalues = self.obj()
            return f
        raise ValueError("torch7_io: unknown type tag %d at byte %d" % (tag, self.p - 4))


def loads(data):
    r = _Reader(bytes(data))
    o = r.obj()
    return o


def load(path):
    """torch.load(path) for binary Torch7 files."""
    with open(path, "rb") as f:
        return loads(f.read())


class _Writer:
    def __init__(self):
        self.out, self.next = [], 1

    def w(self, fmt, *v):
        self.out.append(struct.pack("<" + fmt, *v))

    def string(self, s):
        b = s if isinstance(s, bytes) else s.encode("latin-1")
        self.w("i", len(b))
        self.out.append(b)

    def index(self):
        self.w("i", self.next)
        self.next += 1

    def obj(self, o):
        if o is None:
            self.w("i", 0)
        elif isinstance(o, (bool, np.bool_)):
            self.w("ii", 5, int(o))
        elif isinstance(o, (int, float, np.integer, np.floating)):
            self.w("id", 1, float(o))
        elif isinstance(o, str):
            self.w("i", 2)
            self.string(o)
        elif isinstance(o, (list, tuple)):
            self.obj({float(i + 1): v for i, v in enumerate(o)})
        elif isinstance(o, dict):
            self.w("i", 3)
            self.index()
            self.w("i", len(o))
            for k, v in o.items():
                self.obj(k)
                self.obj(v)
        elif isinstance(o, np.ndarray):
            a = np.ascontiguousarray(o)
            if a.dtype not in _DTYPE_TENSOR:
                raise TypeError("torch7_io.save: no Torch7 tensor type for dtype %s" % a.dtype)
            tname = _DTYPE_TENSOR[a.dtype]
            self.w("i", 4)
            self.index()
            self.string("V 1")
            self.string(tname)
            self.w("i", a.ndim)
            for s in a.shape:
                self.w("q", s)
            for s in a.strides:
                self.w("q", s // a.itemsize)
            self.w("q", 1)
            self.w("i", 4)
            self.index()
            self.string("V 1")
            self.string(_TENSOR_STORAGE[tname])
            self.w("q", a.size)
            self.out.append(a.astype(a.dtype.newbyteorder("<")).tobytes())
        else:
            raise TypeError("torch7_io.save: cannot serialise %r" % type(o))


def dumps(obj):
    w = _Writer()
    w.obj(obj)
    return b"".join(w.out)


def save(path, obj):
    with open(path, "wb") as f:
        f.write(dumps(obj))


def load_calibration(path):
    """The camera calibration table of the radial pipeline (radial/*.cal, version2/rectified_gopro.cal): dict with wImg, hImg,
    K (3x3 float32), distortion (5 float32), bad_image_threshold, sfm (dict of tracker / RANSAC parameters) and, where
    present, rectify.  radial/radial_opticalflow_data.lua:160-170 reads exactly these fields."""
    t = load(path)
    if not isinstance(t, dict) or "K" not in t or "wImg" not in t or "hImg" not in t:
        raise ValueError("%s is not a calibration table (keys: %s)" % (path, sorted(t) if isinstance(t, dict) else type(t)))
    K = np.asarray(t["K"], np.float32)
    if K.shape != (3, 3):
        raise ValueError("%s: K is %s, expected 3x3" % (path, K.shape))
    out = dict(t)
    out["K"], out["wImg"], out["hImg"] = K, int(t["wImg"]), int(t["hImg"])
    if "distortion" in t:
        out["distortion"] = np.asarray(t["distortion"], np.float32)
    return out
