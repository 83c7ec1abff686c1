#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: train images/s of YOLOv8n, 640x640, batch 64 per GPU, bf16, synthetic COCO-80.

A "step" is one pass of the hot path over one batch already resident in HBM:
  Yolov8.forward (train) -> v8DetectionLoss (TAL + CIoU + DFL + BCE, on device) -> backward -> [grad all-reduce] -> AdamW -> zero_grad
Nothing is skipped inside the timed region.  N > 1: one process per GPU -- `python bench.py --gpus N` starts the N ranks itself
(torch.distributed.run on 127.0.0.1; under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE instead) -- batch sharded by image
(weak scaling: 64 images per GPU), SUM all-reduce of gradients over RCCL overlapped with the backward segments.

Prints ONE JSON line on rank 0 (contract in the task statement), extended with
  roofline     -- the DOMINANT KERNEL of the step (largest summed launch time; conv_p2_kernel on config 2, conv_gemm_kernel on
                  config 5): algorithmic bytes (HBM-bound) or flop (MFMA-bound, by the kernel's own arithmetic intensity against
                  the ridge) of its launches / their HIP-event durations, measured live on the engine's stream in extra untimed
                  steps after the timed region (yolosharp_amd/roofline.py); `kernels` lists every convolution kernel the same way
  cpu_baseline -- the oracle (ATen-CPU restatement of the reference, NOT TorchSharp) timed on this host's cores on bounded
                  samples, rank 0 / N=1 only: train step (B=64, the headline batch), predict = forward + NMS, NMS boxes/s
  nms          -- secondary metric: NMS boxes/s on [64, 84, 8400] (candidates entering greedy NMS per second)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def synth_labels(B, nc, seed=1, kmax=16):
    rng = np.random.default_rng(seed)
    bi, cl, bb = [], [], []
    for b in range(B):
        k = int(rng.integers(1, kmax + 1))
        wh = rng.uniform(0.03, 0.6, (k, 2))
        c = wh / 2 + rng.uniform(0, 1, (k, 2)) * (1 - wh)
        bi.append(np.full(k, b, np.float32)); cl.append(rng.integers(0, nc, k).astype(np.float32))
        bb.append(np.concatenate([c, wh], 1).astype(np.float32))
    return np.concatenate(bi), np.concatenate(cl), np.concatenate(bb)


def synth_masks(bi, bb, B, mh, mw):
    """Overlap-encoded instance masks [B, H/4, W/4] (YoloDataset.cs:265-267): each label's box-inscribed ellipse painted
    with its 1-based per-image index, later labels on top."""
    masks = np.zeros((B, mh, mw), np.float32)
    yy, xx = np.meshgrid(np.arange(mh, dtype=np.float32) + 0.5, np.arange(mw, dtype=np.float32) + 0.5, indexing="ij")
    per = [0] * B
    for j in range(len(bi)):
        b = int(bi[j]); per[b] += 1
        cx, cy, w, h = bb[j] * np.array([mw, mh, mw, mh], np.float32)
        inside = ((xx - cx) / max(w / 2, 1e-3)) ** 2 + ((yy - cy) / max(h / 2, 1e-3)) ** 2 <= 1.0
        masks[b][inside] = per[b]
    return masks


def cpu_baseline(nc, H, W, sample_b=64, nms_pred=None):
    """Oracle (port) on the host cores: train step (forward + loss + backward + AdamW, fp32), predict (eval forward + decode +
    NMS, the reference's Detector.ImagePredict path, Detector.cs:27-72) and NMS alone (Ops.cs:239-371 restated in oracle/nms_ref.c)."""
    import ctypes as C
    import torch
    from oracle import yolo_oracle as O
    from yolosharp_amd import build
    torch.manual_seed(0)
    ref = O.Yolov8(nc=nc, size="n").train()
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-4, weight_decay=5e-4)
    crit = O.v8DetectionLoss(nc)
    x = torch.rand(sample_b, 3, H, W)
    batch = O.synthetic_batch(sample_b, H, W, nc, seed=1)
    times = []
    for it in range(2):                # bounded sample: one warm-up step + one timed step of the full batch (~20-40 s on the GPU box's host)
        t0 = time.perf_counter()
        _, preds = ref(x)
        loss, _ = crit(preds, batch)
        opt.zero_grad(); loss.sum().backward(); opt.step()
        times.append(time.perf_counter() - t0)
    t = times[-1]
    out = {"value": round(sample_b / t, 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"oracle/yolo_oracle.py (ATen-CPU restatement, not TorchSharp) YOLOv8n fp32 train step, B={sample_b} {H}x{W} (the headline batch), one step after 1 warm-up step"}
    # ---- predict: eval forward + decode (ATen, all cores) then NMS (plain C, one core) on one image, conf 0.25 / iou 0.45
    lib = C.CDLL(build.build_oracle())
    vp = lambda a: a.ctypes.data_as(C.c_void_p)

    def nms_host(p, conf, iou):
        Bn, Cn, An = p.shape
        rows = np.zeros((Bn, 300, 6), np.float32); keep = np.zeros((Bn, 300), np.int64); cnt = np.zeros(Bn, np.int32)
        t0 = time.perf_counter()
        rc = lib.ys_oracle_nms(vp(p), Bn, Cn, An, C.c_float(conf), C.c_float(iou), 300, 0, 30000, 7680, vp(rows), vp(keep), vp(cnt))
        assert rc == 0
        return time.perf_counter() - t0, int(cnt.sum())
    ref.eval()
    x1 = torch.rand(1, 3, H, W)
    tp = []
    with torch.no_grad():
        for it in range(4):
            t0 = time.perf_counter()
            pred, _ = ref(x1)
            p = np.ascontiguousarray((pred["boxes"] if isinstance(pred, dict) else pred).numpy(), np.float32)
            nms_host(p, 0.25, 0.45)
            tp.append(time.perf_counter() - t0)
    out["predict"] = {"value": round(1.0 / min(tp[1:]), 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                      "sample": f"oracle eval forward + decode (ATen-CPU) + oracle/nms_ref.c, 1x3x{H}x{W}, best of 3 after 1 warm-up (Detector.cs:27-72 path)"}
    if nms_pred is not None:   # the same synthetic [64, 84, 8400] tensor the device NMS line is quoted on
        tn, kept = [], 0
        for it in range(3):
            dt, kept = nms_host(nms_pred.copy(), 0.25, 0.45)
            tn.append(dt)
        ncand = int((nms_pred[:, 4:].max(1) > 0.25).sum())
        out["nms"] = {"value": round(ncand / min(tn), 1), "unit": "boxes/s", "cores": 1, "kind": "port", "ms": round(min(tn) * 1e3, 2),
                      "sample": f"oracle/nms_ref.c (Ops.cs:239-371 + torchvision greedy NMS restated in C, single thread) on {list(nms_pred.shape)}, {ncand} candidates, {kept} kept, best of 3"}
    return out


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv, probe=False):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (torch.distributed.run, one process per GPU,
    rendezvous on 127.0.0.1) and pass their exit status on.  Under a launcher (WORLD_SIZE set) this is never reached."""
    import subprocess
    if not probe:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            raise SystemExit(f"bench.py: --gpus {n} but this node shows {have} GPU(s); refusing to report a {n}-GPU number from fewer devices")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes on this driver)
    env["YS_BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def spawn_probe(backend="torch", expect=0):
    """CPU self-test of the --gpus N launch path (tests/test_dist.py): every rank joins a gloo group, a SUM all-reduce of ones counts
    the ranks, rank 0 prints one JSON line.  No GPU, no engine.  --dist-backend c takes the control-plane steps of the C-ABI path as well:
    rank 0 makes the (here: stand-in) RCCL unique id, broadcast_object_list ships it, every rank checks it received the 128 bytes
    ys_dist_init expects.  The line is checked like the real one: a rank count that differs from --gpus is a non-zero exit."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.ones(1)
    dist.all_reduce(t)
    uid_ok = None
    if backend == "c":
        uid = [bytes(range(128)) if dist.get_rank() == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ok = torch.tensor([1.0 if (isinstance(uid[0], bytes) and len(uid[0]) == 128 and uid[0] == bytes(range(128))) else 0.0])
        dist.all_reduce(ok)
        uid_ok = int(ok.item()) == dist.get_world_size()
    ranks = int(t.item())
    if dist.get_rank() == 0:
        line = {"probe": True, "n_gpus": dist.get_world_size(), "rccl_ranks": ranks, "spawned": os.environ.get("YS_BENCH_SPAWNED") == "1"}
        if backend == "c":
            line["dist_backend"] = "c"; line["uid_broadcast_ok"] = uid_ok
        print(json.dumps(line), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    check_rank_count(ranks, expect or ranks)
    if uid_ok is False:
        raise SystemExit("bench.py: the unique-id broadcast of the C-ABI path did not reach every rank")


def check_rank_count(ranks, n_gpus):
    """A multi-rank line is only a line of record when as many ranks took part in the collective as --gpus promised (round-5 verdict item 9)."""
    if ranks != n_gpus:
        sys.stderr.write(f"bench.py: {ranks} rank(s) took part in the all-reduce but --gpus says {n_gpus}: not a {n_gpus}-GPU measurement\n")
        raise SystemExit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)     # >= 50 timed steps after 10 warm-ups
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--size", default="n")
    ap.add_argument("--imgsz", type=int, default=640, help="square input size (BASELINE configs: 640; config 5 shape: 1280)")
    ap.add_argument("--family", type=int, default=8, choices=[8, 11], help="graph family (8 = YOLOv8, 11 = YOLOv11); default = BASELINE config 2")
    ap.add_argument("--task", default="detect", choices=["detect", "segment", "obb", "pose"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"], help="fp8 = bf16 storage + fp8 MFMA forward / dgrad convolutions (BASELINE config 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-nms", action="store_true")
    ap.add_argument("--no-infer", action="store_true", help="skip the eval-forward secondary metric (profiling runs)")
    ap.add_argument("--force-dist", action="store_true", help="take the torch.distributed (RCCL) code path even with WORLD_SIZE=1 (self-test)")
    ap.add_argument("--dist-backend", default="torch", choices=["torch", "c"],
                    help="gradient exchange of the N > 1 step: torch = torch.distributed all-reduce (RCCL) driven from yolosharp_amd/dist.py; "
                         "c = the library's own ys_dist_init / ys_model_backward_allreduce (RCCL dlopen'ed inside libyolosharp_hip.so: the path a C# host under "
                         "Utils/Amp.cs:260-286 would call); torch.distributed then only carries the rendezvous (unique id broadcast, barriers, timing reductions)")
    ap.add_argument("--lib", default="", help="TRIAGE ONLY: load another build of the library (ablation / timeline variants under build/); the JSON line then carries \"triage_lib\"")
    ap.add_argument("--dump-launches", default="", help="write a per-launch CSV (class,label,us) of the profiled conv launches (triage)")
    ap.add_argument("--spawn-probe", action="store_true", help="self-test of the --gpus N launch path on CPU (gloo): no GPU work")
    ap.add_argument("--cpu-batch", type=int, default=64, help="batch of the cpu_baseline train sample (SURVEY 8d: 64; smaller = faster)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: the bench contract is `bench.py --gpus N`, so start the N ranks here (one process per GPU)
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:], probe=args.spawn_probe))
    if args.spawn_probe:
        return spawn_probe(args.dist_backend, args.gpus)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: launcher started {world} rank(s) but --gpus says {args.gpus}; reporting n_gpus={world}", file=sys.stderr)
    distributed = world > 1 or args.force_dist
    if rank != 0:   # only rank 0 owns stdout (one JSON line); library chatter of the other ranks must not follow it in the merged stream
        try:
            os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
        except Exception:
            pass
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    c_dist = distributed and args.dist_backend == "c"
    rdev = torch.device("cpu") if c_dist else dev     # where the control-plane tensors (timing reductions) live
    if distributed:
        import torch.distributed as dist
        if c_dist:
            # the library brings its own RCCL communicator (dlopen'ed librccl.so): torch.distributed only carries the rendezvous, over gloo -- a second RCCL
            # instance (torch's bundled copy) in the same process is exactly what a C# host would not have
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        if os.environ.get("YS_BENCH_NO_PG_BARRIER") != "1":   # (triage switch)
            dist.barrier()   # the communicator (its streams and buffers) exists before the engine creates its own streams: without it the
                             # engine's two streams ended up serialised (11.6 instead of 10.3 ms/step at one rank)

    from yolosharp_amd import Engine
    from yolosharp_amd.model import Yolov8, Yolov11, Yolov8Segment, Yolov11Segment, v8DetectionLoss, v8SegmentationLoss
    from yolosharp_amd import model as ysm
    from yolosharp_amd import dist as ysd
    from yolosharp_amd.workload import step_work

    nc, H, W, B = {"obb": 15, "pose": 1}.get(args.task, 80), args.imgsz, args.imgsz, args.batch   # DOTA-15 / COCO-person class counts
    stream = None
    if distributed:
        # a stream of its own (not the legacy default stream, whose implicit synchronisation with every blocking stream serialises the
        # engine's side streams): torch's current stream for the whole run, so RCCL collectives order against the engine's kernels
        torch.cuda.set_stream(torch.cuda.Stream(dev))
        stream = torch.cuda.current_stream(dev).cuda_stream
    eng = Engine(local_rank, stream=stream, lib_path=args.lib or None)   # N>1: run on torch's stream so RCCL orders against our kernels
    if os.environ.get("BENCH_WG_FORCE"):                 # TRIAGE ONLY (plan sweeps): "th,tws,per_cu" for every conv_wgrad_tr plan of the process; stamps the line
        import ctypes as _C
        eng.lib.ys_debug_wgrad_force.argtypes = [_C.c_int] * 3
        eng.lib.ys_debug_wgrad_force(*[int(v) for v in os.environ["BENCH_WG_FORCE"].split(",")])
    if not eng.is_device_build:                      # the measured thing is the hipcc-built gfx950 library, never the test interpreter
        raise SystemExit("bench.py: libyolosharp_hip.so is not a device build")
    if c_dist:
        # the library's own communicator, BEFORE the model creates its streams (hardware-queue assignment follows creation order): rank 0 creates
        # the RCCL unique id, torch.distributed ships it (any host channel would do), every rank joins
        uid = [eng.dist_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        eng.dist_init(rank, world, uid[0])
    seg = args.task == "segment"
    Model = {(8, False): Yolov8, (11, False): Yolov11, (8, True): Yolov8Segment, (11, True): Yolov11Segment}[(args.family, seg)]
    if args.task in ("obb", "pose"):
        Model = getattr(ysm, f"Yolov{args.family}{args.task.capitalize()}")
    model = Model(eng, nc=nc, size=args.size, height=H, width=W, max_batch=B, dtype=args.dtype)
    model.init_weights(2)
    model.train()
    crit = v8SegmentationLoss(model) if seg else {"obb": ysm.v8OBBLoss, "pose": ysm.v8PoseLoss}.get(args.task, v8DetectionLoss)(model)
    headline = args.family == 8 and not seg          # BASELINE.json metric/config (YOLOv8 detect)
    rng = np.random.default_rng(0 + rank)
    images = rng.random((B, 3, H, W), dtype=np.float32)
    bi, cl, bb = synth_labels(B, nc, seed=1 + rank)
    d_img = eng.to_device(images)
    if args.task == "obb":       # oriented labels: the same boxes with an angle in [-pi/4, 3pi/4) (Head.cs:429 range)
        bb = np.concatenate([bb, (rng.random((len(bi), 1), dtype=np.float32) - 0.25) * np.float32(np.pi)], 1).astype(np.float32)
    d_lab = (eng.to_device(bi), eng.to_device(cl), eng.to_device(bb), len(bi))
    if args.task == "pose":      # 17 x 3 COCO keypoints scattered in each box, visibility 0 / 1 / 2
        kxy = bb[:, None, :2] + (rng.random((len(bi), 17, 2), dtype=np.float32) - 0.5) * bb[:, None, 2:4]
        kv = rng.integers(0, 3, (len(bi), 17, 1)).astype(np.float32)
        d_lab = d_lab + (eng.to_device(np.ascontiguousarray(np.concatenate([np.clip(kxy, 0, 1), kv], 2), np.float32)),)
    if seg:
        d_lab = d_lab + (eng.to_device(synth_masks(bi, bb, B, H // 4, W // 4)),)
    lr0 = round(0.002 * 5 / (4 + nc), 6)
    lrs = [lr0, lr0, lr0]
    gptr, gn = model.grad_buffer()
    seg_ranges = [model.segment_grad_range(s) for s in range(model.num_segments())]
    sync = None
    if distributed and not c_dist:
        flat = ysd.device_view(gptr.value, gn, dev)
        sync = ysd.GradSync(flat, seg_ranges, model=model,
                            force_collective=args.force_dist)   # --force-dist: the one-rank group still issues its (identity) all-reduces

    def local_step():
        model.forward_device(d_img, B)
        crit.forward_device(*d_lab)
        model.backward()
        model.adamw_step(lrs)
        model.zero_grad()

    def c_step():       # the C# host's step: forward, criterion, segmented backward with each segment's all-reduce issued by the library, AdamW
        model.forward_device(d_img, B)
        crit.forward_device(*d_lab)
        model.backward_allreduce()
        model.adamw_step(lrs)
        model.zero_grad()

    def step():
        if c_dist:
            c_step()
        elif sync is not None:
            ysd.train_step_dp(model, crit, sync, d_img, B, d_lab, lrs)
        else:
            local_step()

    def barrier():
        if distributed:
            dist.barrier()
        eng.synchronize()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter() - t0     # the host has ENQUEUED every step (diagnostic: a value close to `elapsed` means the GPU waits for the host)
    barrier()
    elapsed = time.perf_counter() - t0
    rccl_ranks = 1
    dist_info = None
    items = crit.read()[1]                          # loss items of the last TIMED step (before the untimed diagnostic steps below)
    if distributed:
        own = elapsed
        t = torch.tensor([elapsed], device=rdev, dtype=torch.float64)
        tmin = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        elapsed = float(t.item())
        ones = torch.ones(1, device=rdev, dtype=torch.float32)
        dist.all_reduce(ones)                      # SUM over the process group (RCCL, or gloo with --dist-backend c): how many ranks really took part
        rccl_ranks = int(ones.item())
        # what the exchange costs this step: the same ranks run the LOCAL step (no all-reduce) for a short timed region right after, between the same
        # barriers -- exposed all-reduce time = distributed step - local step (max over ranks of each)
        n_loc = max(5, min(args.steps, 20))
        for _ in range(3):
            local_step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(n_loc):
            local_step()
        barrier()
        tl = torch.tensor([(time.perf_counter() - t1) / n_loc], device=rdev, dtype=torch.float64)
        dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        ms_local = float(tl.item()) * 1e3
        dist_info = {"backend": "c (ys_dist_* in libyolosharp_hip.so)" if c_dist else "torch.distributed (nccl = RCCL)",
                     "rank_ms_per_step_min": round(float(tmin.item()) / args.steps * 1e3, 3), "rank_ms_per_step_max": round(elapsed / args.steps * 1e3, 3),
                     "local_step_ms": round(ms_local, 3), "allreduce_exposed_ms": round(elapsed / args.steps * 1e3 - ms_local, 3),
                     "segment_allreduce_bytes": [int(c) * 4 for _, c in seg_ranges], "allreduce_bytes_per_step": int(sum(c for _, c in seg_ranges)) * 4,
                     "note": "allreduce_exposed_ms = ms_per_step (max over ranks) - the same ranks' local step (forward + criterion + backward + AdamW, no exchange) timed right after; "
                             "segments in backward order (head, neck, late backbone, stem): each all-reduce is issued when its segment's gradients are complete and overlaps the next segment"}
    ms = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed

    out = None
    if rank == 0:
        es = 2 if args.dtype in ("bf16", "fp8") else 4      # fp8 mode stores activations in bf16 (f8.hip)
        wk = step_work(args.size, nc, H, W, es) if headline else None
        # ---- per-kernel roofline, HIP events on the engine stream around every convolution launch, untimed extra steps
        from yolosharp_amd import roofline as RL
        import tempfile
        steps_prof = 2
        model.set_overlap(False)   # per-kernel durations: every kernel alone on the chip (the timed region ran with the weight-gradient stream on)
        eng.kernel_profile(True)
        for _ in range(steps_prof):
            local_step()        # rank 0 only: no collective may be issued here (the other ranks are already at the final barrier)
        eng.synchronize()
        model.set_overlap(True)
        n_ig, ms_ig = eng.kernel_profile_read("conv_igemm")
        n_wg, ms_wg = eng.kernel_profile_read("conv_wgrad")
        dump = args.dump_launches or os.path.join(tempfile.gettempdir(), f"ys_launches_{os.getpid()}.csv")
        eng.kernel_profile_dump(dump)
        eng.kernel_profile(False)
        agg = RL.per_kernel(dump, steps_prof, es)
        if not args.dump_launches:
            os.remove(dump)
        dom = max(agg, key=lambda k: agg[k]["ms"])
        roofline = RL.roofline_of(dom, agg[dom])
        roofline["what"] = ("dominant kernel = largest summed launch time of the step (profile steps run with the second weight-gradient stream off, every kernel alone on the chip); achieved = its launches' algorithmic "
                            + ("bytes" if roofline["bound"] == "hbm" else "flop") + " (launch geometry: input tensor + output tensor once, SURVEY 8d) / their HIP-event durations")
        # HBM bytes per launch of that kernel from the committed PMC passes (rocprofv3 cannot run inside the timed process): only when
        # the file was measured on THIS source tree and configuration
        traffic, traffic_note = None, None
        cfg_key = f"YOLOv{args.family}{args.size}{'-' + args.task if args.task != 'detect' else ''} B={B} {H}x{W} {args.dtype}"
        import glob
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True)   # newest round first
        for tf in tfiles:
            tname = "profiles/" + os.path.basename(tf)
            try:
                ent = json.load(open(tf)).get("configs", {}).get(cfg_key)
                if ent is None:
                    traffic_note = traffic_note or "no PMC passes committed for this configuration"
                elif ent.get("source_sha") != RL.source_sha(ROOT):
                    traffic_note = f"{tname} was measured on source {ent.get('source_sha')}, this tree is {RL.source_sha(ROOT)}: not reported"
                    break
                else:
                    traffic = ent["kernels"][dom.split("<")[0]]["hbm_bytes_per_launch_corrected"]
                    traffic_note = None
                    break
            except Exception as e:
                traffic_note = f"{tname} unreadable ({type(e).__name__})"
        roofline["traffic"] = traffic
        if traffic_note:
            roofline["traffic_note"] = traffic_note
        roofline["kernels"] = {k: {f: v for f, v in RL.roofline_of(k, a).items() if f in ("bound", "achieved", "unit", "frac", "launches_per_step", "avg_launch_ms", "kernel_ms_per_step")}
                               for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
        roofline["class_ms_per_step"] = {"conv_igemm": round(ms_ig / steps_prof, 3), "conv_wgrad": round(ms_wg / steps_prof, 3)}
        # whole step: SURVEY 8d totals (headline graph: yolosharp_amd/workload.py; other graphs: the launches actually made -- conv units only)
        step_bytes = wk["train_bytes"] * B if headline else sum(a["bytes"] for a in agg.values())
        step_flop = wk["train_flop"] * B if headline else sum(a["flop"] for a in agg.values())
        roofline["step_algorithmic_GBps"] = round(step_bytes / (ms * 1e-3) / 1e9, 1)
        roofline["step_frac"] = round(step_bytes / (ms * 1e-3) / 1e9 / RL.HBM_PEAK_GBS, 4)
        roofline["step_TFLOPs"] = round(step_flop / (ms * 1e-3) / 1e12, 2)
        roofline["step_mfma_frac"] = round(step_flop / (ms * 1e-3) / 1e12 / RL.MFMA_PEAK_TF["fp8" if args.dtype == "fp8" else ("bf16" if args.dtype == "bf16" else "f32")], 4)
        gname = f"YOLOv{args.family}{args.size}" + {"segment": "-seg", "obb": "-obb", "pose": "-pose"}.get(args.task, "")
        out = {"metric": f"train images/sec {gname} {W}x{H} bs={B}/GPU", "value": round(value, 2), "unit": "images/s",
               "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"{gname} {args.task} train step (fwd+loss+bwd+AdamW), {B}x3x{H}x{W} per GPU, {'COCO-80' if nc == 80 else str(nc) + '-class'} synthetic labels" + {"segment": " + instance masks", "obb": " (oriented)", "pose": " + 17x3 keypoints"}.get(args.task, ""),
                          "global_batch": B * world, "parallelism": f"dp{world}"},
               "loss_items": [round(float(v), 5) for v in items], "roofline": roofline,
               "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 3)}
        if dist_info is not None:
            out["dist"] = dist_info
        if args.lib:
            out["triage_lib"] = args.lib          # not the product library: never a bench line of record
        if os.environ.get("BENCH_WG_FORCE"):
            out["triage_wg_force"] = os.environ["BENCH_WG_FORCE"]
        # ---- secondary metric: inference images/s = eval forward (BN folded into the conv epilogues) + Detect decode
        model.eval()
        for _ in range(0 if args.no_infer else 3):
            model.forward_device(d_img, B)
        eng.synchronize()
        t1 = time.perf_counter()
        n_inf = 10
        for _ in range(0 if args.no_infer else n_inf):
            model.forward_device(d_img, B)
        eng.synchronize()
        t_inf = (time.perf_counter() - t1) / n_inf
        out["infer"] = None if args.no_infer else {"images_per_s": round(B / t_inf, 1), "ms_per_batch": round(t_inf * 1e3, 3), "batch": B,
                        "what": "eval forward + decode to pred [B,4+nc(+nm),A], inputs resident in HBM"}
        model.train()
        # ---- secondary metric: NMS boxes/s on [64, 84, 8400]
        nms_pred_host = None
        if not args.no_nms and nc == 80:      # the NMS line is quoted on the COCO-80 shape
            prng = np.random.default_rng(3)
            A = 8400
            wh = prng.uniform(0.03, 0.6, (64, 2, A)) * 640; c = prng.uniform(0, 640, (64, 2, A))
            sc = 1 / (1 + np.exp(-prng.normal(-3, 1.5, (64, nc, A))))
            sc[:, :, prng.random(A) < 0.92] *= 0.05                      # ~8 % of anchors pass conf 0.25
            pred = np.concatenate([c, wh, sc], 1).astype(np.float32)
            nms_pred_host = pred
            ncand = int(((pred[:, 4:].max(1)) > 0.25).sum())
            d_pred = eng.malloc(pred.nbytes)
            d_rows = eng.malloc(64 * 300 * 6 * 4); d_keep = eng.malloc(64 * 300 * 8); d_cnt = eng.malloc(64 * 4)
            import ctypes as C
            times = []
            for it in range(6):
                # restore the xywh input (NMS converts boxes in place), untimed
                eng.lib.ys_memcpy_h2d(eng.ctx, d_pred, pred.ctypes.data_as(C.c_void_p), pred.nbytes)
                eng.synchronize()
                t1 = time.perf_counter()
                eng.nms_device(d_pred, 64, 84, A, 0.25, 0.45, 300, 0, d_rows, d_keep, d_cnt)
                eng.synchronize()
                times.append(time.perf_counter() - t1)
            tn = float(np.median(times[1:]))
            out["nms"] = {"boxes_per_s": round(ncand / tn, 1), "candidates": ncand, "ms": round(tn * 1e3, 3),
                          "shape": [64, 84, A], "conf": 0.25, "iou": 0.45}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nc, H, W, sample_b=args.cpu_batch, nms_pred=nms_pred_host)
        else:
            out["cpu_baseline"] = None
        # the JSON line must be the LAST line of rank 0's stdout: libraries (RCCL prints "Librccl path : ..." through C stdio,
        # block-buffered when piped) would otherwise land after it at exit.  Flush C stdio first, print, then close fd 1.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
        try:
            os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
        except Exception:
            pass
    if distributed:
        dist.barrier()
        if c_dist:
            eng.dist_destroy()
        dist.destroy_process_group()
        check_rank_count(rccl_ranks, args.gpus)     # every rank: the line above stays readable, the exit status says it is not an N-GPU measurement


if __name__ == "__main__":
    main()
