"""Which phase of the training step is not run-to-run deterministic: one model, fixed weights (no optimizer step), the same batch N times; per trial the head outputs, the
loss items, the criterion's gradients and every parameter gradient are hashed and compared with the first trial.  usage: det_phase.py [trials] [batch]"""
import os, sys, hashlib, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from yolosharp_amd import Engine
from yolosharp_amd.model import Yolov8, v8DetectionLoss
from bench import synth_labels

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
eng = Engine(0)
rng = np.random.default_rng(0)
images = rng.random((B, 3, 640, 640), dtype=np.float32)
bi, cl, bb = synth_labels(B, 80, seed=1)
d_img = eng.to_device(images)
d_lab = (eng.to_device(bi), eng.to_device(cl), eng.to_device(bb), len(bi))
def make():
    m = Yolov8(eng, nc=80, size=os.environ.get("DET_SIZE", "n"), height=640, width=640, max_batch=B, dtype=os.environ.get("DET_DTYPE", "bf16"))
    m.init_weights(2); m.train(); m.set_overlap(os.environ.get("DET_OVERLAP", "1") == "1")
    return m, v8DetectionLoss(m)
m, crit = make()
NEW = os.environ.get("DET_NEW", "0") == "1"      # a new model per trial (first step on recycled device memory)
PRE = int(os.environ.get("DET_PRE", "0"))        # untimed full steps (with AdamW) before the hashed one, on a new model
lr0 = round(0.002 * 5 / 84, 6)
h = lambda a: hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()[:8]
ref = None
bad = collections.Counter()
for t in range(trials):
    if NEW and t:
        m.close(); m, crit = make()
    for _ in range(PRE):
        m.forward_device(d_img, B); crit.forward_device(*d_lab); m.backward(); m.adamw_step([lr0] * 3); m.zero_grad()
    m.forward_device(d_img, B); eng.synchronize(); m._batch = B
    cur = {"out.boxes": h(m.get_output("boxes")), "out.scores": h(m.get_output("scores"))}
    crit.forward_device(*d_lab); eng.synchronize()
    cur["loss.items"] = h(crit.read()[1]); cur["loss.dboxes"] = h(m.get_output("dboxes")); cur["loss.dscores"] = h(m.get_output("dscores"))
    m.backward(); eng.synchronize()
    g = m.grads()
    for k, v in g.items():
        cur["grad." + k] = h(v)
    m.zero_grad()
    if ref is None:
        ref = cur
        continue
    diff = [k for k in cur if cur[k] != ref[k]]
    if diff:
        nan = [k for k in diff if k.startswith("grad.") and not np.isfinite(g[k[5:]]).all()]
        print(f"trial {t}: {len(diff)} differ; first {diff[:6]}{' ... NaN in ' + str(nan[:3]) if nan else ''}", flush=True)
        for k in diff:
            bad[k] += 1
print("trials", trials, "keys that ever differed from trial 0:", len(bad))
for k, v in bad.most_common(40):
    print(f"  {v:3d}  {k}")
