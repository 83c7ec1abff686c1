"""Cut a small golden fixture out of the reference's only shipped weight file (data, not source):
the container header + the first 6 tensors of YoloSharpDemo/Assets/PreTrainedModels/Yolov5n.bin, and what the reference's
own reader (Utils/Lib.cs:9-54) must produce for them.  Run in the dev container (needs /root/reference); the outputs are committed."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from yolosharp_amd import weights_bin as W

SRC = "/root/reference/YoloSharpDemo/Assets/PreTrainedModels/Yolov5n.bin"
HERE = os.path.dirname(os.path.abspath(__file__))
K = 6
with open(SRC, "rb") as f:
    total = W._leb_read(f)
    f.seek(0)
    items = list(W.iter_bin(f, limit=K))
    end = f.tell()
    f.seek(0)
    prefix = f.read(end)
with open(SRC, "rb") as f:
    n_all = sum(1 for _ in W.iter_bin(f))
    rest = f.read()
open(os.path.join(HERE, "yolov5n_prefix.bin"), "wb").write(prefix)
meta = {"tensor_count": total, "file_bytes": os.path.getsize(SRC), "consumed_exactly": n_all == total and rest == b"",
        "tensors": [{"name": n, "scalar_type": c, "shape": list(a.shape),
                     "sum_f64": float(np.asarray(a if c != W.BFLOAT16 else W.bf16_to_f32(a), np.float64).sum()),
                     "first": [float(v) for v in np.asarray(a if c != W.BFLOAT16 else W.bf16_to_f32(a), np.float64).reshape(-1)[:4]]}
                    for n, c, a in items]}
json.dump(meta, open(os.path.join(HERE, "yolov5n_prefix.json"), "w"), indent=1)
print(json.dumps(meta)[:600], len(prefix))
