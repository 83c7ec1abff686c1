"""bf16 parity of YOLOv8n at 640x640 with PRODUCTION kernel routing (round-3 verdict, weak item 2): tests/conftest.py lowers the size
gates of the blocked-GEMM / fp8 kernels for the whole suite so that oracle-sized shapes reach them, which means the suite's full-size
tests do not run the kernels bench.py times.  This module runs the step in a fresh process without those overrides
(tests/workers/prod_routing_worker.py) and compares it with the oracle -- the rounding-matched one (tests/bf16_ref.py) for the
tight statements, the plain fp32 one for the north-star-style statement.  Reference: Models/Yolo.cs:92-134, Utils/Loss.cs:411-484."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "workers", "prod_routing_worker.py")


def run_worker(tmp_path, B, H, W, emu=False, f32=False, **kw):
    out = str(tmp_path / "prod.npz")
    env = {k: v for k, v in os.environ.items() if not k.startswith("YS_")}
    cmd = [sys.executable, WORKER, out, str(B), str(H), str(W)] + (["emu"] if emu else []) + (["f32"] if f32 else []) + ["%s=%s" % kv for kv in kw.items()]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1700)
    assert r.returncode == 0, r.stdout[-3000:]
    return np.load(out, allow_pickle=False)


def summarize(d):
    """Distance of the engine's step from both oracles: loss items, head outputs per element (in units of the per-element bound of
    tests/bf16_ref.py), parameter gradients (cosine per tensor and overall, norm ratio)."""
    from bf16_ref import elem_bound
    out = {"items": d["items"], "r_items": d["r_items"], "f_items": d["f_items"]}
    for k in [k for k in ("boxes", "scores", "mask_coefficient", "proto") if k in d.files]:
        for tag in ("r", "f"):
            ref = d[tag + "_" + k].astype(np.float64)
            r = np.abs(d[k] - ref) / elem_bound(ref)
            out["%s_%s" % (k, tag)] = (float((r > 1).mean()), float((r > 8).mean()), float(r.max()),
                                       float(np.sqrt(((d[k] - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean())))
    if "inf" in d.files:
        # eval forward (decoded predictions [B, 4 + nc (+ nm), A]): box rows in pixels, class rows as probabilities -- (max |diff|, rms diff) against both oracles
        for tag in ("r", "f"):
            a, b = d["inf"].astype(np.float64), d[tag + "_inf"].astype(np.float64)
            out["inf_box_" + tag] = (float(np.abs(a[:, :4] - b[:, :4]).max()), float(np.sqrt(((a[:, :4] - b[:, :4]) ** 2).mean())))
            out["inf_cls_" + tag] = (float(np.abs(a[:, 4:84] - b[:, 4:84]).max()), float(np.sqrt(((a[:, 4:84] - b[:, 4:84]) ** 2).mean())))
    names = [k[4:] for k in d.files if k.startswith("e_g_") and "r_g_" + k[4:] in d.files]
    for tag in ("r", "f"):
        num = da = db = 0.0
        per = []
        for n in names:
            a, b = d["e_g_" + n].astype(np.float64).ravel(), d[tag + "_g_" + n].astype(np.float64).ravel()
            num += float(a @ b); da += float(a @ a); db += float(b @ b)
            per.append((float(a @ b) / np.sqrt(float(a @ a) * float(b @ b) + 1e-300), n, float(np.sqrt(b @ b))))
        # analytically-zero gradients (SPPF.cv1.bn.bias: a per-channel shift commutes with the max pools and is removed by cv2's batch
        # statistics) are rounding noise in both implementations: direction is only defined for tensors of non-negligible norm
        med = float(np.median([p[2] for p in per]))
        per = sorted((c, n) for c, n, nb in per if nb > 1e-2 * med and d["e_g_" + n].size >= 64)
        out["grad_" + tag] = (num / np.sqrt(da * db), np.sqrt(da / db), per[:3])
    out["n_grads"] = len(names)
    return out


def compare(d, big):
    s = summarize(d)
    dump = os.path.join(ROOT, "gpurun_out")
    if big and os.path.isdir(dump):
        with open(os.path.join(dump, "prod_routing_summary.txt"), "w") as f:
            for k, v in s.items():
                f.write("%s: %s\n" % (k, v))
    assert s["n_grads"] > 150
    if big:
        _check_eval(s)
    if not big:
        # 64 x 64, B = 2: the P5 BatchNorms normalise over 8 values per channel, so one flipped bf16 value moves whole channels -- this
        # size only keeps the worker and the comparison code exercised in the CPU suite
        assert np.allclose(s["items"], s["r_items"], rtol=5e-2), (s["items"], s["r_items"])
        assert s["grad_r"][0] > 0.95, s["grad_r"]
        return
    # loss items: against the rounding-matched oracle, and against the plain fp32 oracle
    assert np.allclose(s["items"], s["r_items"], rtol=ITEMS_R), (s["items"], s["r_items"])
    assert np.allclose(s["items"], s["f_items"], rtol=ITEMS_F), (s["items"], s["f_items"])
    # head outputs against the rounding-matched oracle: rms distance, and the tail of the per-element ratio
    for k in ("boxes", "scores"):
        frac1, frac8, worst, rms = s[k + "_r"]
        assert rms < HEAD_RMS and frac8 < HEAD_FRAC8, (k, s[k + "_r"])
    # every parameter gradient: direction and size
    cos, ratio, per = s["grad_r"]
    assert cos > GRAD_COS and abs(ratio - 1.0) < GRAD_NORM, s["grad_r"]
    assert per[0][0] > GRAD_COS_MIN, per


# Thresholds of the 640 x 640, B = 8 case.  Calibration run on the MI355X (round 4, production routing, against the rounding-matched
# oracle / the plain fp32 oracle): loss items within 6e-4 / 2.8e-3 / 1.6e-4 (box / cls / dfl) and 8e-4 / 1.9e-3 / 1e-3; head outputs 6.6-6.8 %
# rms (10.7-11.1 % against fp32 -- i.e. bf16 storage itself moves the outputs of this randomly initialised 22-layer graph by ~10 %, which
# is why the tight bf16 statements live at the block level, tests/test_blocks.py); parameter gradients: overall cosine 0.9974, norm ratio
# 0.9972, worst tensor 0.896 (model.22.cv2.2.0.bn.weight).  Margins of 2-4x on top.
# Round 5: HEAD_FRAC8 1.0 -> 0.35 (measured 0.225 / 0.239: the bound was vacuous), GRAD_COS_MIN 0.8 -> 0.88 (measured worst tensor 0.896 - 0.937).
ITEMS_R, ITEMS_F = 1e-2, 1e-2
HEAD_RMS, HEAD_FRAC8 = 0.15, 0.35
GRAD_COS, GRAD_NORM, GRAD_COS_MIN = 0.99, 1.5e-2, 0.88


@pytest.mark.gpu
def test_v8n_640_b8_bf16_production_routing(tmp_path):
    d = run_worker(tmp_path, 8, 640, 640)
    labels = [str(l).split(",")[1] for l in d["labels"] if str(l).startswith("conv_igemm")]
    # production gates: P5 layers (20 x 20 x 8 = 3200 pixels, >= 128 channels) on the blocked-GEMM kernel, narrow layers on the patch kernel
    assert any(l.startswith("gemm ") for l in labels) and any(l.startswith("p2") for l in labels), labels[:5]
    compare(d, big=True)


def _relerr(a, b):
    """Per-element distance of tests/test_model.py: max |a - b| / (|b| + rms(b))."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float((np.abs(a - b) / (np.abs(b) + max(float(np.sqrt((b * b).mean())), 1e-6))).max())


def _dump(name, s, extra=()):
    dump = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(dump):
        with open(os.path.join(dump, name), "w") as f:
            for k, v in s.items():
                f.write("%s: %s\n" % (k, v))
            for line in extra:
                f.write(line + "\n")


def _check_eval(s):
    """Eval forward (folded BatchNorm: no batch statistics, nothing chaotic) under production routing: decoded boxes within 0.15 px of the rounding-matched oracle and 0.2 px
    of the plain fp32 one, class probabilities within 1e-3 of both -- measured on the MI355X (round 6, all five configurations): 0.025-0.043 px, 1.6e-4 - 2.4e-4."""
    if "inf_box_r" not in s:
        return
    assert s["inf_box_r"][0] < 0.15 and s["inf_box_f"][0] < 0.2, (s["inf_box_r"], s["inf_box_f"])
    assert s["inf_cls_r"][0] < 1e-3 and s["inf_cls_f"][0] < 1e-3, (s["inf_cls_r"], s["inf_cls_f"])


def _conv_labels(d):
    """(class, plan label) of every profiled convolution launch of the step the worker ran."""
    return [tuple(str(l).split(",")[:2]) for l in d["labels"]]


@pytest.mark.gpu
def test_v8n_640_b64_production_routing(tmp_path):
    """The HEADLINE point (BASELINE config 2: YOLOv8n, B = 64, 640 x 640 -- the exact plan set bench.py times: tile plans, grouped-grid residency and persistent tile
    walks depend on M = B * H * W) against the oracle, in its own process = production routing: the bf16 engine against the rounding-matched oracle (items, head
    outputs, gradient direction and size), and the fp32 engine against the plain oracle IN DOUBLE at the north-star tolerance (items and head logits 1e-3, every
    parameter gradient per element 2e-3).  Reference step: Utils/Amp.cs:260-286, loss Utils/Loss.cs:411-477.  The two oracle passes cost ~25 s each on the GPU box's host cores."""
    d = run_worker(tmp_path, 64, 640, 640, f32=True)
    s = summarize(d)
    # bf16, production routing, against the rounding-matched oracle: items, head outputs (the B = 8 test's bounds), gradients
    assert np.allclose(s["items"], s["r_items"], rtol=ITEMS_R), (s["items"], s["r_items"])
    for k in ("boxes", "scores"):
        frac1, frac8, worst, rms = s[k + "_r"]
        assert rms < HEAD_RMS and frac8 < HEAD_FRAC8, (k, s[k + "_r"])
    cos, ratio, per = s["grad_r"]
    assert cos > GRAD_COS and abs(ratio - 1.0) < GRAD_NORM, s["grad_r"]
    assert per[0][0] > GRAD_COS_MIN, per
    # fp32 engine against the plain oracle in double: north-star tolerance on the items AND on the head logits, per element
    assert np.allclose(d["items32"], d["f_items"], rtol=1e-3, atol=1e-5), (d["items32"], d["f_items"])
    head32 = {k: _relerr(d["e32_" + k], d["f_" + k]) for k in ("boxes", "scores")}
    names = [k[6:] for k in d.files if k.startswith("e32_g_") and "f_g_" + k[6:] in d.files]
    assert len(names) > 150
    gscale = max(float(np.abs(d["f_g_" + n]).max()) for n in names)
    ratios = []
    for n in names:                            # the criterion of test_model.py::test_full_resolution_parity_f32 (B = 4), at the headline batch, against the oracle in double
        a, b = d["e32_g_" + n].astype(np.float64), d["f_g_" + n].astype(np.float64)
        ratios.append((float(np.abs(a - b).max() / (np.abs(b).max() + 1e-3 * gscale)), n))
    ratios.sort(reverse=True)
    _dump("prod_routing_b64_summary.txt", s, ["f32 head logits, per-element relerr: %s" % (head32,),
                                               "f32 gradient tensors by max |a - b| / (max |b| + 1e-3 gscale), worst first: %s" % (ratios[:8],)])
    assert max(head32.values()) < 1e-3, head32
    _check_eval(s)
    # Round 6: 2e-3 of each tensor's maximum for EVERY tensor (the round-5 run measured 4.9e-4 on the worst one, model.9.cv1.conv.weight, against the oracle in double;
    # the 2.5e-2 / "at most 4 tensors >= 1e-2" of round 5 described the FLOAT oracle's own summation error and would have passed a wrong tile column)
    assert ratios[0][0] < 2e-3, ratios[:4]


# ---- the other BASELINE configurations under production routing (round-5 verdict 1b / 5b: tests/conftest.py lowers the routing gates for the whole suite, so
# test_configs.py pushes 20 x 20 / 40 x 40 maps through the 16 x 16 halo tiles and 1x1 / 32-channel layers through fp8; these runs have no YS_* in their environment).
# Graph family / size / task / dtype go to the worker; the bf16 engine is compared with the rounding-matched oracle (tests/bf16_ref.py: Conv, Bottleneck, Detect / Segment
# towers, C2PSA attention, Proto), the loss items also with the plain fp32 oracle.  Thresholds: CFG[...] = (items vs matched, items vs fp32, head rms, overall gradient
# cosine, |norm ratio - 1|, worst-tensor cosine); measured values next to each (gpurun_out/prod_routing_<tag>_summary.txt of the calibration run, round 6).
CFG = {
    # config 3 per-GPU shape: YOLOv8s B = 32.  Measured (round 6, MI355X): items 6e-4 / 8e-4 / 7e-4 from the matched oracle, head rms 0.073-0.074, gradient cosine 0.99921,
    # norm ratio 1.0007, worst tensor 0.954 (model.22.cv2.2.0.conv.weight)
    "v8s_b32": (dict(B=32, H=640, W=640, family=8, size="s"), (5e-3, 1.5e-2, 0.15, 0.998, 5e-3, 0.90)),
    # config 4 graph: YOLOv11m-seg, B = 4 (the oracle's per-image mask loop and the C2PSA attention in fp32 on host cores bound the batch).  damp = 0.25: see the worker.
    # Measured: items 1.7e-3 / 4.7e-3 / 1.6e-3 / 5e-5, head rms 0.070-0.090, cosine 0.99218, norm ratio 0.9995, worst tensor 0.923 (model.22.m.0.m.1.cv2.bn.weight)
    # (undamped: cosine 0.955, worst 0.778 -- and the two ORACLES then differ by more than that)
    "v11m_seg_b4": (dict(B=4, H=640, W=640, family=11, size="m", task="segment", damp=0.25), (1.5e-2, 3e-2, 0.15, 0.985, 1e-2, 0.85)),
    # config 5 graph and resolution: YOLOv8x 1280 x 1280, B = 2, bf16 and fp8; damp = 0.25.  bf16 measured: items 4e-4 / 8e-4 / 1.2e-3, head rms 0.052, cosine 0.99964,
    # norm ratio 1.0008, worst tensor 0.926 (model.22.cv2.0.1.conv.weight) (undamped: cosine 0.776, worst 0.43)
    "v8x_1280_b2": (dict(B=2, H=1280, W=1280, family=8, size="x", damp=0.25), (5e-3, 1e-2, 0.10, 0.999, 5e-3, 0.85)),
    # fp8 against the BF16-matched oracle (no e4m3 / e5m2 operand rounding in it): the distance is the fp8 mode's own -- items 9e-4 / 7.7e-2 / 3.4e-3, head rms 0.39
    # (e4m3 operands carry ~2^-4 relative rounding per element, i.e. percents per layer output, through 60 convolutions), cosine 0.9940, norm ratio 1.077, worst tensor 0.459
    # (model.21.m.1.cv1.bn.weight).  Per-layer statements against quantised-operand fp32 are in tests/test_fp8.py.
    "v8x_1280_b2_fp8": (dict(B=2, H=1280, W=1280, family=8, size="x", dtype="fp8", damp=0.25), (1.2e-1, 1.2e-1, 0.60, 0.98, 0.12, 0.30)),
}
# launch labels that must occur (class, prefix): the kernels bench.py times on that configuration
WANT = {
    "v8s_b32": [("conv_igemm", "gemm "), ("conv_igemm", "p2"), ("conv_igemm", "halo "), ("conv_wgrad", "wgemm "), ("conv_wgrad", "wgrad_tr ")],
    "v11m_seg_b4": [("conv_igemm", "gemm "), ("conv_igemm", "p2"), ("conv_wgrad", "wgemm ")],
    "v8x_1280_b2": [("conv_igemm", "gemm "), ("conv_igemm", "halo "), ("conv_igemm", "p2"), ("conv_wgrad", "wgemm ")],
    "v8x_1280_b2_fp8": [("conv_igemm", "gemmf8 "), ("conv_igemm", "p2"), ("conv_wgrad", "wgemm ")],
}


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(CFG))
def test_other_configs_production_routing(tmp_path, tag):
    a, (it_r, it_f, head_rms, gcos, gnorm, gmin) = CFG[tag]
    a = dict(a)
    B, H, W = a.pop("B"), a.pop("H"), a.pop("W")
    d = run_worker(tmp_path, B, H, W, **a)
    labels = _conv_labels(d)
    s = summarize(d)
    kinds = sorted(set((c, l.split(" ")[0]) for c, l in labels))
    _dump("prod_routing_%s_summary.txt" % tag, s, ["launch kinds: %s" % (kinds,)] + ["%s,%s" % cl for cl in sorted(set(labels))])
    for c, pre in WANT[tag]:
        assert any(cl == c and l.startswith(pre) for cl, l in labels), (c, pre, kinds)
    if tag == "v8x_1280_b2":
        assert any(l.startswith("halo ") and "tile16x16" in l for _, l in labels) and any(l.startswith("halo ") and "tile8x16" in l for _, l in labels), kinds
    assert np.all(np.isfinite(s["items"]))
    _check_eval(s)
    assert np.allclose(s["items"], s["r_items"], rtol=it_r), (s["items"], s["r_items"])
    assert np.allclose(s["items"], s["f_items"], rtol=it_f), (s["items"], s["f_items"])
    for k in [k for k in ("boxes", "scores", "mask_coefficient", "proto") if k + "_r" in s]:
        assert s[k + "_r"][3] < head_rms, (k, s[k + "_r"])
    cos, ratio, per = s["grad_r"]
    assert cos > gcos and abs(ratio - 1.0) < gnorm, s["grad_r"]
    assert per[0][0] > gmin, per


DET = {
    "v8s_b32": dict(B=32, H=640, W=640, family=8, size="s"),
    "v11m_seg_b4": dict(B=4, H=640, W=640, family=11, size="m", task="segment"),
    "v8x_1280_b2_fp8": dict(B=2, H=1280, W=1280, family=8, size="x", dtype="fp8"),
}


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(DET))
def test_fresh_model_first_step_is_deterministic_production_routing(tmp_path, tag):
    """Three freshly created models, same seed: the first training step of each (fp8: the second as well -- the first runs the bf16 kernels) must agree bit for bit in the
    loss items, the head outputs and EVERY gradient tensor, under production routing (round-5 verdict 5c: the round-3 race in stem_wgrad_kernel only ever showed on a
    model's first step; test_model.py covers YOLOv8n)."""
    a = dict(DET[tag])
    B, H, W = a.pop("B"), a.pop("H"), a.pop("W")
    d = run_worker(tmp_path, B, H, W, mode="det", **a)
    names = [str(n) for n in d["names"]]
    assert len(names) > 150
    diff = [n for n, p0, p1, p2 in zip(names, d["per0"], d["per1"], d["per2"]) if not (p0 == p1 == p2)]
    assert not diff, diff[:8]
    assert str(d["hash0"]) == str(d["hash1"]) == str(d["hash2"]), (d["items0"], d["items1"], d["items2"])


def test_worker_runs_on_the_interpreter(tmp_path):
    """The same script at 64 x 64 on the test interpreter (CPU suite): keeps the worker and the comparison code exercised."""
    d = run_worker(tmp_path, 2, 64, 64, emu=True)
    compare(d, big=False)
