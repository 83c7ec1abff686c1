"""The criterion alone (config-2 shapes), N calls -- run under rocprofv3 --kernel-trace --stats to get its kernels' durations.  usage: crit_time.py [lib]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from yolosharp_amd import Engine
from yolosharp_amd.model import Yolov8, v8DetectionLoss
from bench import synth_labels
lib = sys.argv[1] if len(sys.argv) > 1 else None
eng = Engine(0, lib_path=lib) if lib else Engine(0)
B = 64
m = Yolov8(eng, nc=80, size="n", height=640, width=640, max_batch=B, dtype="bf16")
m.init_weights(2); m.train()
crit = v8DetectionLoss(m)
img = eng.to_device(np.random.default_rng(0).random((B, 3, 640, 640), dtype=np.float32))
bi, cl, bb = synth_labels(B, 80, seed=1)
lab = (eng.to_device(bi), eng.to_device(cl), eng.to_device(bb), len(bi))
m.forward_device(img, B)
for _ in range(20):
    crit.forward_device(*lab)
eng.synchronize()
print("labels", len(bi))
