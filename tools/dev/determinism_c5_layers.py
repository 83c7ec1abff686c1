"""Triage: which BatchNorm layers of the config-5 graph differ between two fresh model instances after ONE training forward
(running_mean is the per-layer fingerprint of the batch statistics)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from yolosharp_amd import Engine
from yolosharp_amd.model import Yolov8
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] not in ("", "-") else None
eng = Engine(0, lib_path=lib) if lib else Engine(0)
B, H, W, nc = 16, 1280, 1280, 80
x = np.random.default_rng(51).random((B, 3, H, W), dtype=np.float32)
sds = []
for r in range(3):
    m = Yolov8(eng, nc=nc, size="x", height=H, width=W, max_batch=B, dtype="bf16")
    m.init_weights(7); m.train()
    m.forward(x, fetch=False)
    sd = m.state_dict()
    sds.append({k: np.array(v, copy=True) for k, v in sd.items() if k.endswith("running_mean")})
    m.close()
for r in (1, 2):
    bad = [k for k in sds[0] if not np.array_equal(sds[0][k], sds[r][k])]
    print("round %d: %d of %d BN layers differ" % (r, len(bad), len(sds[0])))
    for k in bad[:12]:
        d = np.nonzero(sds[0][k] != sds[r][k])[0]
        print("   %s: C=%d, %d channels differ, first %s, max rel %.2e" % (k, sds[0][k].size, d.size, d[:8], float(np.abs(sds[0][k] - sds[r][k]).max() / (np.abs(sds[0][k]).max() + 1e-30))))
