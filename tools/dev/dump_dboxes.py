import sys
sys.path.insert(0, ".")
import numpy as np
from oracle import yolo_oracle as O
from yolosharp_amd import Engine
from yolosharp_amd.model import Yolov8, v8DetectionLoss
eng = Engine()
B, H, W, nc = 16, 320, 320, 80
x = np.random.default_rng(0).random((B, 3, H, W), dtype=np.float32)
batch = {k: v.numpy() for k, v in O.synthetic_batch(B, H, W, nc, seed=1, kmax=8).items()}
m = Yolov8(eng, nc=nc, size="n", height=H, width=W, max_batch=B, dtype=sys.argv[2])
m.init_weights(11); m.train()
_, preds = m.forward(x)
_, items = v8DetectionLoss(m)(None, batch)
np.save(sys.argv[1], m.get_output("dboxes")[:4])
