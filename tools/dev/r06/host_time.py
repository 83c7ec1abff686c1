"""Host-side cost of enqueueing one training step (config 2 by default): wall time of each C-ABI call with the GPU idle at the start of the step
(so nothing measured is the host waiting for the device), against the GPU time of the same step.  usage: host_time.py [size] [batch] [imgsz]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from yolosharp_amd import Engine
from yolosharp_amd.model import Yolov8, v8DetectionLoss
size = sys.argv[1] if len(sys.argv) > 1 else "n"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H = W = int(sys.argv[3]) if len(sys.argv) > 3 else 640
eng = Engine(0)
m = Yolov8(eng, nc=80, size=size, height=H, width=W, max_batch=B, dtype="bf16")
m.init_weights(2); m.train()
crit = v8DetectionLoss(m)
rng = np.random.default_rng(0)
img = eng.to_device(rng.random((B, 3, H, W), dtype=np.float32))
n = B * 8
bi = np.sort(rng.integers(0, B, n)).astype(np.float32); cl = rng.integers(0, 80, n).astype(np.float32)
bb = np.concatenate([rng.random((n, 2), dtype=np.float32) * 0.8 + 0.1, rng.random((n, 2), dtype=np.float32) * 0.3 + 0.02], 1).astype(np.float32)
lab = (eng.to_device(bi), eng.to_device(cl), eng.to_device(bb), n)
lrs = [1e-4] * 3
calls = [("forward", lambda: m.forward_device(img, B)), ("criterion", lambda: crit.forward_device(*lab)), ("backward", lambda: m.backward()),
         ("adamw", lambda: m.adamw_step(lrs)), ("zero_grad", lambda: m.zero_grad())]
for _ in range(5):
    for _, f in calls: f()
eng.synchronize()
N = 20
acc = {k: 0.0 for k, _ in calls}; tot_host = 0.0; tot_gpu = 0.0
for _ in range(N):
    eng.synchronize()
    t0 = time.perf_counter()
    for k, f in calls:
        t1 = time.perf_counter(); f(); acc[k] += time.perf_counter() - t1
    t2 = time.perf_counter()
    eng.synchronize()
    t3 = time.perf_counter()
    tot_host += t2 - t0; tot_gpu += t3 - t0
print("host enqueue per step: %.3f ms   (step wall with sync: %.3f ms)" % (tot_host / N * 1e3, tot_gpu / N * 1e3))
for k, _ in calls: print("  %-10s %.3f ms" % (k, acc[k] / N * 1e3))
# free-running (no per-step sync): the bench's regime
eng.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    for _, f in calls: f()
t1 = time.perf_counter(); eng.synchronize(); t2 = time.perf_counter()
print("free-running: host %.3f ms/step, wall %.3f ms/step" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
