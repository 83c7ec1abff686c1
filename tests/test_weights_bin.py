"""`.bin` weight container (Utils/Lib.cs:9-54 reader, YoloBaseTaskModel.cs:470-490,538-559 writer).
Golden: the header + first 6 tensors of the reference's shipped Yolov5n.bin (tests/golden/make_bin_fixture.py)."""
import io
import json
import os

import numpy as np
import pytest

from conftest import BACKENDS
from yolosharp_amd import weights_bin as W

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_reference_file_prefix():
    meta = json.load(open(os.path.join(GOLD, "yolov5n_prefix.json")))
    assert meta["consumed_exactly"] and meta["tensor_count"] == 348
    with open(os.path.join(GOLD, "yolov5n_prefix.bin"), "rb") as f:
        assert W._leb_read(f) == 348
        f.seek(0)
        got = list(W.iter_bin(f, limit=len(meta["tensors"])))
        assert f.read() == b""                                   # the fixture ends exactly after the 6th tensor
    for (name, code, a), t in zip(got, meta["tensors"]):
        assert (name, code, list(a.shape)) == (t["name"], t["scalar_type"], t["shape"])
        v = np.asarray(a, np.float64)
        assert v.sum() == t["sum_f64"] and list(v.reshape(-1)[:4]) == t["first"]
    assert got[0][0] == "model.0.conv.weight" and got[0][1] == W.FLOAT16


def test_leb128_and_net_string():
    for v in (0, 1, 127, 128, 300, 16383, 16384, 3774197, 2 ** 40):
        b = io.BytesIO(); W._leb_write(b, v); b.seek(0)
        assert W._leb_read(b) == v
    b = io.BytesIO(); W._leb_write(b, 300)
    assert b.getvalue() == bytes([0xAC, 0x02])                   # 300 = 0b10_0101100 -> AC 02 (BinaryWriter 7-bit int)
    with pytest.raises(NotImplementedError):
        W._leb_write(io.BytesIO(), -1)


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_write_read_round_trip(tmp_path, dtype):
    rng = np.random.default_rng(0)
    sd = {"model.0.conv.weight": rng.standard_normal((16, 3, 3, 3)).astype(np.float32),
          "model.22.one2one_cv2.0.0.conv.weight": np.ones((4, 4, 1, 1), np.float32),     # skipped by SaveWeight
          "model.0.bn.running_var": rng.random(16).astype(np.float32) + 0.5,
          "model.0.bn.num_batches_tracked": np.asarray(7, np.int64),
          "model.22.dfl.conv.weight": np.arange(16, dtype=np.float32).reshape(1, 16, 1, 1),
          "ünïcode.name": np.zeros((0,), np.float32)}
    p = str(tmp_path / "w.bin")
    W.write_bin(p, sd, dtype)
    back, code = W.read_bin(p)
    assert list(back) == [k for k in sd if "one2one" not in k] and code == W._CODE[dtype]
    assert back["model.0.bn.num_batches_tracked"].dtype == np.int64 and int(back["model.0.bn.num_batches_tracked"]) == 7
    tol = {"f32": 0.0, "f16": 2 ** -10, "bf16": 2 ** -8}[dtype]
    for k in ("model.0.conv.weight", "model.0.bn.running_var", "model.22.dfl.conv.weight"):
        assert back[k].shape == sd[k].shape and np.all(np.abs(back[k] - sd[k]) <= tol * np.abs(sd[k]) + 1e-12)
    # bf16 rounding is round-to-nearest-even on the bit pattern
    x = np.array([1.0, 1.00390625, 1.01171875, -3.140625, 65504.0], np.float32)
    assert np.array_equal(W.bf16_to_f32(W.f32_to_bf16(x)), np.array([1.0, 1.0, 1.015625, -3.140625, 65536.0], np.float32))


@pytest.mark.parametrize("backend", BACKENDS)
def test_engine_save_load(backend, engine, tmp_path):
    """SaveWeight / LoadModel through the engine: TorchSharp state_dict order, bit-exact fp32 round trip."""
    from yolosharp_amd.model import Yolov8
    m = Yolov8(engine, nc=80, size="n", height=64, width=64, max_batch=1, dtype="f32")
    m.init_weights(5)
    p = str(tmp_path / "last.bin")
    W.save_from(m, p, "f32")
    sd, code = W.read_bin(p)
    assert code == W.FLOAT32 and list(sd) == [n for n, _, _ in m.tensor_info()]
    assert sd["model.0.bn.num_batches_tracked"].dtype == np.int64
    m2 = Yolov8(engine, nc=80, size="n", height=64, width=64, max_batch=1, dtype="f32")
    names = W.load_into(m2, p)
    assert len(names) == len(sd)
    a, b = m.state_dict(), m2.state_dict()
    assert all(np.array_equal(a[k], b[k]) for k in a)
    with pytest.raises(KeyError):
        W.write_bin(p, {"not.a.tensor": np.zeros(3, np.float32)}); W.load_into(m2, p)
    m.close(); m2.close()
