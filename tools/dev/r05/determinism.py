"""Run-to-run determinism of the training step: the same model / seed / batch built and stepped N times in one process, per option set; prints the distinct loss-item
triples and how often each came out.  usage: determinism.py [trials] [batch] [steps]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from yolosharp_amd import Engine
from yolosharp_amd.model import Yolov8, v8DetectionLoss
from bench import synth_labels

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
size = os.environ.get("DET_SIZE", "n")
eng = Engine(0)
rng = np.random.default_rng(0)
images = rng.random((B, 3, 640, 640), dtype=np.float32)
bi, cl, bb = synth_labels(B, 80, seed=1)
d_img = eng.to_device(images)
d_lab = (eng.to_device(bi), eng.to_device(cl), eng.to_device(bb), len(bi))
lr0 = round(0.002 * 5 / 84, 6)

def one(overlap):
    m = Yolov8(eng, nc=80, size=size, height=640, width=640, max_batch=B, dtype="bf16")
    m.init_weights(2); m.train(); m.set_overlap(overlap)
    crit = v8DetectionLoss(m)
    for _ in range(steps):
        m.forward_device(d_img, B); crit.forward_device(*d_lab); m.backward(); m.adamw_step([lr0] * 3); m.zero_grad()
    eng.synchronize()
    it = tuple(float(v) for v in crit.read()[1])
    del crit, m
    return it

sets = [("production", {}, True), ("overlap off", {}, False), ("GEMM_HALO=0", {"GEMM_HALO": 0}, True), ("BN_ATOMIC=0", {"BN_ATOMIC": 0}, True),
        ("halo0 atom0", {"GEMM_HALO": 0, "BN_ATOMIC": 0}, True), ("halo0 atom0 overlap off", {"GEMM_HALO": 0, "BN_ATOMIC": 0}, False)]
only = os.environ.get("DET_ONLY")
for name, opts, ov in sets:
    if only and only not in name:
        continue
    import contextlib
    with (eng.options(**opts) if hasattr(eng, 'options') else contextlib.nullcontext()):
        c = collections.Counter(one(ov) for _ in range(trials))
    print(f"{name:28s} distinct={len(c)}  " + "  ".join(f"{k[0]:.5f}/{k[1]:.2f}x{v}" for k, v in c.most_common()), flush=True)
