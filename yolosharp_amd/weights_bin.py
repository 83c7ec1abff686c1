"""YoloSharp `.bin` weight container.

Reader  <- Utils/Lib.cs:9-54  (`Lib.LoadModel`): LEB128 tensor count; per tensor a .NET `BinaryReader.ReadString` name
           (7-bit-encoded byte length + UTF-8), LEB128 TorchSharp `ScalarType`, LEB128 ndim, LEB128 dims, raw little-endian data.
Writer  <- Models/YoloBaseTaskModel.cs:470-490 (`SaveWeight`: state_dict order, keys containing "one2one" skipped) and
           `Encode` (:538-559) + TorchSharp `Tensor.Save` (same per-tensor header as the reader consumes).

Host-side only (numpy): the engine keeps fp32 master weights, so every floating tensor is converted to fp32 on load and to
the requested storage type on save.  state_dict names and order are the engine's `ys_model_tensor_info` listing, which is
TorchSharp's (tests/test_model.py::test_state_dict_surface).
"""
import collections
import io

import numpy as np

# TorchSharp ScalarType codes (torch c10::ScalarType)
BYTE, INT8, INT16, INT32, INT64, FLOAT16, FLOAT32, FLOAT64, BOOL, BFLOAT16 = 0, 1, 2, 3, 4, 5, 6, 7, 11, 15
_NP = {BYTE: np.uint8, INT8: np.int8, INT16: np.int16, INT32: np.int32, INT64: np.int64, FLOAT16: np.float16,
       FLOAT32: np.float32, FLOAT64: np.float64, BOOL: np.bool_, BFLOAT16: np.uint16}
_CODE = {"f32": FLOAT32, "float32": FLOAT32, "f16": FLOAT16, "float16": FLOAT16, "bf16": BFLOAT16, "bfloat16": BFLOAT16}


def _leb_read(f):
    num, shift = 0, 0
    while True:
        b = f.read(1)
        if not b:
            raise EOFError("truncated LEB128 value")
        num |= (b[0] & 0x7F) << shift
        if not b[0] & 0x80:
            return num
        shift += 7


def _leb_write(f, value):
    if value < 0:
        raise NotImplementedError("LEB128 encoding of negative numbers")      # YoloBaseTaskModel.cs:540-543
    while True:
        b = value & 0x7F
        value >>= 7
        if value == 0:
            f.write(bytes([b]))
            return
        f.write(bytes([b | 0x80]))


def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(x):
    """Round-to-nearest-even, NaN preserved (what torch's .to(bfloat16) does)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    out = ((u + r) >> 16).astype(np.uint16)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    out[nan] = ((u[nan] >> 16) | 0x40).astype(np.uint16)
    return out


def iter_bin(f, limit=None):
    """Yield (name, scalar_type_code, ndarray in the stored type; bf16 as uint16 bit patterns) in file order."""
    count = _leb_read(f)
    for i in range(count if limit is None else min(count, limit)):
        n = _leb_read(f)
        name = f.read(n).decode("utf-8")
        code = _leb_read(f)
        if code not in _NP:
            raise ValueError(f"{name}: unsupported ScalarType {code}")
        shape = [_leb_read(f) for _ in range(_leb_read(f))]
        dt = np.dtype(_NP[code]).newbyteorder("<")
        nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        raw = f.read(nbytes)
        if len(raw) != nbytes:
            raise EOFError(f"{name}: truncated tensor data")
        yield name, code, np.frombuffer(raw, dt).reshape(shape).copy()


def read_bin(path, as_float32=True):
    """Lib.LoadModel: name -> ndarray (floating tensors as fp32 when as_float32), in file order; also returns the model
    dtype code = the first tensor's (Lib.cs:37-40)."""
    sd, model_code = collections.OrderedDict(), None
    with open(path, "rb") as f:
        for name, code, a in iter_bin(f):
            if model_code is None:
                model_code = code
            if as_float32 and code == BFLOAT16:
                a = bf16_to_f32(a)
            elif as_float32 and code in (FLOAT16, FLOAT64):
                a = a.astype(np.float32)
            sd[name] = a
        if f.read(1):
            raise ValueError("trailing bytes after the last tensor")
    return sd, model_code


def write_bin(path, state_dict, dtype="f32"):
    """SaveWeight: every entry of `state_dict` (name -> ndarray) in order, except names containing "one2one".  Floating
    tensors are stored as `dtype`; integer tensors (num_batches_tracked) keep int64 like TorchSharp's buffers."""
    code = _CODE[dtype]
    items = [(k, v) for k, v in state_dict.items() if "one2one" not in k]
    buf = io.BytesIO()
    _leb_write(buf, len(items))
    for name, v in items:
        a = np.asarray(v)
        nb = name.encode("utf-8")
        _leb_write(buf, len(nb))
        buf.write(nb)
        if a.dtype.kind in "iu" or name.endswith("num_batches_tracked"):
            a, c = a.astype("<i8"), INT64
        elif code == BFLOAT16:
            a, c = f32_to_bf16(a.astype(np.float32)).astype("<u2"), BFLOAT16
        else:
            a, c = a.astype(np.dtype(_NP[code]).newbyteorder("<")), code
        _leb_write(buf, c)
        _leb_write(buf, a.ndim)
        for d in a.shape:
            _leb_write(buf, int(d))
        buf.write(a.tobytes())
    with open(path, "wb") as f:
        f.write(buf.getvalue())


def load_into(model, path, strict=True):
    """YoloBaseTaskModel.LoadModel (:27-114) for an engine model: names are identical, shapes are checked by the engine."""
    sd, _ = read_bin(path)
    want = {n: s for n, s, _ in model.tensor_info()}
    out = {}
    for name, a in sd.items():
        if name not in want:
            if strict:
                raise KeyError(f"{name} is not a tensor of this model")
            continue
        out[name] = np.asarray(a, np.float32).reshape(want[name])     # num_batches_tracked: int64 scalar -> fp32 [1]
    model.load_state_dict(out, strict=strict)
    return list(out)


def save_from(model, path, dtype="f32"):
    """SaveWeight for an engine model (state_dict order = TorchSharp's: parameters, then buffers)."""
    sd = collections.OrderedDict()
    for name, a in model.state_dict().items():
        sd[name] = np.asarray(np.rint(a).reshape(()), np.int64) if name.endswith("num_batches_tracked") else a
    write_bin(path, sd, dtype)
