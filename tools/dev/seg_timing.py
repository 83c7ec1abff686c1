"""Triage: step time of the YOLOv8n B=64 train step with the backward pass as one call, as four synchronous segments and as four
asynchronous segments (no process group, engine on its own stream)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as bench_mod
from yolosharp_amd import Engine
from yolosharp_amd.model import Yolov8, v8DetectionLoss
import torch
PG = "--pg" in sys.argv; TS = "--torch-stream" in sys.argv
if PG:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, **({"device_id": torch.device("cuda", 0)} if "--devid" in sys.argv else {}))
    dist.barrier()
stream = None
if TS:
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream(0))
    stream = torch.cuda.current_stream(0).cuda_stream
eng = Engine(0, stream=stream) if TS else Engine(0)
print("process group", PG, "| engine on a torch stream", TS)
B, H, W, nc = 64, 640, 640, 80
model = Yolov8(eng, nc=nc, size="n", height=H, width=W, max_batch=B, dtype="bf16")
model.init_weights(2); model.train()
crit = v8DetectionLoss(model)
rng = np.random.default_rng(0)
d_img = eng.to_device(rng.random((B, 3, H, W), dtype=np.float32))
bi, cl, bb = bench_mod.synth_labels(B, nc, seed=1)
d_lab = (eng.to_device(bi), eng.to_device(cl), eng.to_device(bb), len(bi))
lrs = [1e-4] * 3
from yolosharp_amd import dist as ysd
sync = None
if "--gradsync" in sys.argv:
    gptr, gn = model.grad_buffer()
    flat = ysd.device_view(gptr.value, gn, torch.device("cuda", 0))
    sync = ysd.GradSync(flat, [model.segment_grad_range(s) for s in range(model.num_segments())], model=(model if "--fence" in sys.argv else None))
def step(mode):
    if mode == "dp":
        ysd.train_step_dp(model, crit, sync, d_img, B, d_lab, lrs); return
    model.forward_device(d_img, B); crit.forward_device(*d_lab)
    if mode == "whole":
        model.backward()
    else:
        for s in range(model.num_segments()):
            (model.backward_segment_async if mode == "async" else model.backward_segment)(s)
    model.adamw_step(lrs); model.zero_grad()
for rep in range(1):
    for mode in (("dp", "whole", "dp") if sync is not None else ("whole", "async")):
        for _ in range(8): step(mode)
        eng.synchronize(); t0 = time.perf_counter()
        for _ in range(30): step(mode)
        eng.synchronize()
        print("%-6s %.3f ms/step" % (mode, (time.perf_counter() - t0) / 30 * 1e3))
