"""Data-parallel path: world_size-2 gloo test of GradSync -- the SUM all-reduce of per-shard gradients equals
the gradient of the single-device global batch when BN statistics are per shard (the reference has no SyncBN), and the
bucketed/segmented path equals one flat all-reduce."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import BACKENDS
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import yolo_oracle as O
    from yolosharp_amd.dist import GradSync
    torch.manual_seed(0)
    nc, H, W, Bl = 4, 64, 64, 2
    ref = O.Yolov8(nc=nc, size="n").train()
    x_all = torch.rand(world * Bl, 3, H, W, generator=torch.Generator().manual_seed(1))
    batch_all = O.synthetic_batch(world * Bl, H, W, nc, seed=2, kmax=3)
    # shard by image
    lo, hi = rank * Bl, (rank + 1) * Bl
    sel = (batch_all["batch_idx"] >= lo) & (batch_all["batch_idx"] < hi)
    batch = {"batch_idx": batch_all["batch_idx"][sel] - lo, "cls": batch_all["cls"][sel], "bboxes": batch_all["bboxes"][sel]}
    _, preds = ref(x_all[lo:hi])
    loss, _ = O.v8DetectionLoss(nc)(preds, batch)
    loss.sum().backward()
    names = [n for n, p in ref.named_parameters() if p.grad is not None]
    flat = torch.cat([dict(ref.named_parameters())[n].grad.reshape(-1) for n in names]).clone()
    local = flat.clone()
    n = flat.numel()
    cuts = [0, n // 5, n // 2, n]
    sync = GradSync(flat, [(cuts[i], cuts[i + 1] - cuts[i]) for i in range(3)])
    for seg in range(3):          # head -> neck -> backbone order of the engine's backward segments
        sync.allreduce_segment(seg)
    sync.wait()
    # reference: gather every rank's local gradient and sum in rank order
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = sum(gathered)
    q.put((rank, bool(torch.allclose(flat, expect, rtol=1e-6, atol=1e-7)), float((flat - expect).abs().max()), float(local.abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, err, mx in res:
        assert ok, (rank, err, mx)
        assert mx > 0


def _engine_worker(rank, world, port, q, emu_path, task="detect"):
    """One rank of the data-parallel step on the interpreter build: ENGINE forward / loss / segmented backward, GradSync's
    bucketed SUM all-reduce overlapped per segment (gloo), AdamW, zero_grad -- dist.train_step_dp, the loop bench.py runs."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import yolo_oracle as O
    from yolosharp_amd import Engine
    from yolosharp_amd import dist as ysd
    from yolosharp_amd.model import Yolov8, Yolov8Pose, v8DetectionLoss, v8PoseLoss
    torch.manual_seed(0)
    nc, H, W, Bl = 4, 64, 64, 2
    pose = task == "pose"      # Pose n: 51-wide towers padded inside -> the flat gradient buffer carries zero rows on every rank
    ref = (O.Yolov8Pose if pose else O.Yolov8)(nc=nc, size="n").train()
    ocrit = O.v8PoseLoss(nc) if pose else O.v8DetectionLoss(nc)
    sd0 = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    x_all = torch.rand(world * Bl, 3, H, W, generator=torch.Generator().manual_seed(1))
    batch_all = O.synthetic_batch(world * Bl, H, W, nc, seed=2, kmax=3)
    if pose:
        batch_all["keypoints"] = O.synthetic_keypoints(batch_all)

    def shard(r):
        lo, hi = r * Bl, (r + 1) * Bl
        sel = (batch_all["batch_idx"] >= lo) & (batch_all["batch_idx"] < hi)
        out = {"batch_idx": batch_all["batch_idx"][sel] - lo, "cls": batch_all["cls"][sel], "bboxes": batch_all["bboxes"][sel]}
        if pose:
            out["keypoints"] = batch_all["keypoints"][sel]
        return x_all[lo:hi], out

    eng = Engine(lib_path=emu_path)
    m = (Yolov8Pose if pose else Yolov8)(eng, nc=nc, size="n", height=H, width=W, max_batch=Bl, dtype="f32")
    m.load_state_dict({k: v.numpy() for k, v in sd0.items()})
    m.train()
    crit = (v8PoseLoss if pose else v8DetectionLoss)(m)
    x, b = shard(rank)
    d_img = eng.to_device(x.numpy())
    nb = {k: v.numpy() for k, v in b.items()}
    d_lab = (eng.to_device(nb["batch_idx"]), eng.to_device(nb["cls"]), eng.to_device(nb["bboxes"]), len(nb["batch_idx"]))
    if pose:
        d_lab = d_lab + (eng.to_device(np.ascontiguousarray(nb["keypoints"], np.float32)),)
    gptr, gn = m.grad_buffer()
    flat = ysd.host_view(gptr.value, gn)
    sync = ysd.GradSync(flat, [m.segment_grad_range(s) for s in range(m.num_segments())])
    # ---- step 1 by hand (the body of train_step_dp up to the optimizer) so the all-reduced gradient can be inspected
    m.zero_grad()
    m.forward_device(d_img, Bl); crit.forward_device(*d_lab)
    for seg in range(m.num_segments()):
        m.backward_segment(seg); eng.synchronize(); sync.allreduce_segment(seg)
    sync.wait()
    got = {k: v.copy() for k, v in m.grads().items()}
    # oracle: global-batch gradient under PER-SHARD BatchNorm statistics = sum over shards of d(loss_shard * B_shard)
    want = None
    for r in range(world):
        ref.load_state_dict(sd0); ref.zero_grad()
        xs, bs = shard(r)
        _, preds = ref(xs)
        loss, _ = ocrit(preds, bs)
        loss.sum().backward()
        g = {n: p.grad.detach().clone() for n, p in ref.named_parameters() if p.grad is not None}
        want = g if want is None else {n: want[n] + g[n] for n in g}
    gscale = max(float(v.abs().max()) for v in want.values())
    worst = max(float(np.abs(got[n] - want[n].numpy()).max()) / (float(want[n].abs().max()) + 1e-3 * gscale) for n in want)
    lrs = [1e-3] * 3
    m.adamw_step(lrs); m.zero_grad()
    # ---- step 2 through the packaged loop
    ysd.train_step_dp(m, crit, sync, d_img, Bl, d_lab, lrs)
    eng.synchronize()
    pptr, pn = m.param_buffer()
    params = ysd.host_view(pptr.value, pn).clone()
    gathered = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(gathered, params)
    same = all(torch.equal(gathered[0], t) for t in gathered[1:])
    q.put((rank, worst, same, bool(torch.isfinite(params).all())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("task", ["detect", "pose"])
def test_train_step_dp_world2_engine_gloo(emu_lib_path, task):
    """the data-parallel step driven end to end by the ENGINE (interpreter build) over gloo, world size 2:
    the segment-overlapped all-reduced gradient equals the oracle's global-batch gradient under per-shard BN, and after AdamW
    both ranks hold bit-identical weights (two steps)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_engine_worker, args=(r, 2, port, q, emu_lib_path, task)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, worst, same, finite in res:
        assert worst < 2e-3, (rank, worst)
        assert same and finite, rank


def test_train_step_dp_world4_engine_gloo(emu_lib_path):
    """The same end-to-end data-parallel step with FOUR ranks (round-2 verdict item 8): global batch 8 sharded 2 per rank, four
    backward segments, all-reduced gradient equal to the oracle's under per-shard BN, bit-identical weights on all ranks.
    Unmeasured on hardware: the development boxes have one GPU (gloo on CPU tensors here)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_engine_worker, args=(r, 4, port, q, emu_lib_path, "detect")) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(4)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, worst, same, finite in res:
        assert worst < 2e-3, (rank, worst)
        assert same and finite, rank


def _bf16_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from yolosharp_amd.dist import GradSync
    g = torch.Generator().manual_seed(10 + rank)
    flat = torch.randn(1000, generator=g)
    local = flat.clone()
    sync = GradSync(flat, [(0, 300), (300, 700)], compress="bf16")
    sync.allreduce_segment(0); sync.allreduce_segment(1); sync.wait()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = sum(t.bfloat16().float() for t in gathered)
    results = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(results, flat)
    q.put((rank, float((flat - expect).abs().max() / expect.abs().max()), all(torch.equal(results[0], t) for t in results[1:])))
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_bf16_exchange_world2_gloo():
    """compress="bf16": the exchanged values are the bf16 roundings, the result is identical on every rank and within bf16
    accumulation rounding of the fp32 sum of those roundings."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bf16_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, same in res:
        assert err < 1e-2 and same, (rank, err, same)


def test_gradsync_single_process_is_identity():
    from yolosharp_amd.dist import GradSync
    g = torch.arange(10, dtype=torch.float32)
    s = GradSync(g, [(0, 4), (4, 6)])
    s.allreduce_segment(0); s.allreduce_segment(1); s.wait()
    assert torch.equal(g, torch.arange(10, dtype=torch.float32))


def test_segment_ranges_cover_flat_buffer(emu_lib_path):
    """The engine's backward segments partition the flat gradient buffer (head, neck, late backbone, stem); the stem -- whose
    all-reduce is the exposed one -- is the smallest (round-2 verdict item 8: <= 1/4 of the buffer)."""
    from yolosharp_amd import Engine
    from yolosharp_amd.model import Yolov8
    eng = Engine(lib_path=emu_lib_path)
    m = Yolov8(eng, nc=80, size="n", height=64, width=64, max_batch=1, dtype="f32")
    ranges = [m.segment_grad_range(s) for s in range(m.num_segments())]
    _, n = m.grad_buffer()
    assert ranges[0][0] == 0 and sum(c for _, c in ranges) == n == m.num_params()
    for (o1, c1), (o2, _) in zip(ranges, ranges[1:]):
        assert o1 + c1 == o2
    assert len(ranges) == 4 and ranges[-1][1] * 4 <= n and ranges[-1][1] == min(c for _, c in ranges), ranges
    m.close()


@pytest.mark.parametrize("backend", BACKENDS)
def test_async_segment_ends_give_the_same_gradients(backend, engine):
    """ys_model_backward_segment_async + ys_model_segment_fence (a segment ends without the main stream waiting for the
    weight-gradient stream; the consumer waits on the segment's events) against the synchronous segments and the one-call
    backward: identical gradients and, after AdamW (which orders itself behind the weight-gradient stream), identical weights."""
    from yolosharp_amd.model import Yolov8, v8DetectionLoss
    from oracle import yolo_oracle as O
    B, H, W, nc = 2, 64, 64, 6
    x = np.random.default_rng(5).random((B, 3, H, W), dtype=np.float32)
    nb = {k: v.numpy() for k, v in O.synthetic_batch(B, H, W, nc, seed=6).items()}
    res = {}
    for mode in ("whole", "sync", "async"):
        m = Yolov8(engine, nc=nc, size="n", height=H, width=W, max_batch=B, dtype="bf16")
        m.init_weights(3); m.train()
        crit = v8DetectionLoss(m)
        for it in range(2):
            m.zero_grad()
            m.forward(x, fetch=False); crit(None, nb)
            if mode == "whole":
                m.backward()
            else:
                for seg in range(m.num_segments()):
                    if mode == "async":
                        m.backward_segment_async(seg); m.segment_fence(seg, 0)     # stream 0: the fence itself is what is exercised
                    else:
                        m.backward_segment(seg)
            grads = {k: v.copy() for k, v in m.grads().items()}                      # get_grad orders itself behind the weight-gradient stream
            m.adamw_step([1e-3] * 3)
        res[mode] = (grads, {k: np.array(v, copy=True) for k, v in m.state_dict().items()})
        m.close()
    for mode in ("sync", "async"):
        for k in res["whole"][0]:
            assert np.array_equal(res["whole"][0][k], res[mode][0][k]), (mode, k)
        for k in res["whole"][1]:
            assert np.array_equal(res["whole"][1][k], res[mode][1][k]), (mode, k)


def test_c_abi_dist_needs_device_build():
    """The interpreter build has no RCCL: the C-level exchange must refuse loudly, never silently skip the all-reduce."""
    import ctypes as C
    from yolosharp_amd import _lib, build
    lib = _lib.load(build.build_emu())
    buf = (C.c_ubyte * 128)()
    assert lib.ys_dist_unique_id(buf) == 4 and b"interpreter" in lib.ys_last_error()


@pytest.mark.gpu
def test_c_abi_rccl_world1():
    """ys_dist_* on a real GPU with a one-rank communicator: the overlapped segmented backward + all-reduce reproduces the
    plain backward bit for bit (SUM over one rank), AdamW runs behind ys_dist_wait, and misuse is reported."""
    from yolosharp_amd import Engine, YsError
    from yolosharp_amd.model import Yolov8, v8DetectionLoss
    eng = Engine(0)
    B, H, W, nc = 4, 128, 128, 80
    x = np.random.default_rng(0).random((B, 3, H, W), dtype=np.float32)
    sys.path.insert(0, ROOT)
    from oracle import yolo_oracle as O
    batch = {k: v.numpy() for k, v in O.synthetic_batch(B, H, W, nc, seed=1, kmax=5).items()}
    m = Yolov8(eng, nc=nc, size="n", height=H, width=W, max_batch=B, dtype="bf16")
    m.init_weights(3); m.train()
    m.forward(x, fetch=False); v8DetectionLoss(m)(None, batch)
    with pytest.raises(YsError):
        m.backward_allreduce()                       # no communicator yet
    m.zero_grad(); m.backward()
    ref = m.grads()
    eng.dist_init(0, 1, eng.dist_unique_id())
    with pytest.raises(YsError):
        eng.dist_init(0, 1, eng.dist_unique_id())    # one communicator per context
    m.forward(x, fetch=False); v8DetectionLoss(m)(None, batch)
    m.zero_grad(); m.backward_allreduce()
    got = m.grads()
    assert all(np.array_equal(ref[k], got[k]) for k in ref)
    m.adamw_step([1e-3] * 3); m.zero_grad()
    eng.synchronize()
    eng.dist_destroy()
    m.close()


@pytest.mark.gpu
def test_bench_force_dist_world1():
    """bench.py's torch.distributed (RCCL) code path -- engine on torch's stream, GradSync over the segmented backward -- executed
    on the GPU box with a one-rank process group; the JSON contract line must come out and agree with a plain run's loss."""
    import json
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--no-nms", "--no-infer"]
    outs = []
    for extra in (["--force-dist"], [], ["--force-dist", "--dist-backend", "c"]):
        r = subprocess.run(cmd + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    a, b, c = outs
    assert a["n_gpus"] == 1 and a["scaling"] == "weak" and a["config"]["parallelism"] == "dp1" and a["value"] > 0
    assert a["roofline"]["bound"] == "hbm" and 0 < a["roofline"]["frac"] < 1
    assert np.allclose(a["loss_items"], b["loss_items"], rtol=1e-5)          # SUM over one rank: identical training trajectory
    # round 5: the same step with the exchange driven by the library itself (ys_dist_init / ys_model_backward_allreduce: the C# host's path), and the
    # first-run diagnostics of a multi-rank line (exposed all-reduce time, per-segment bytes, slowest / fastest rank)
    assert np.allclose(c["loss_items"], b["loss_items"], rtol=1e-5)
    assert "dist" not in b
    for o, tag in ((a, "torch"), (c, "c (")):
        d = o["dist"]
        assert d["backend"].startswith(tag) and len(d["segment_allreduce_bytes"]) == 4 and d["allreduce_bytes_per_step"] == sum(d["segment_allreduce_bytes"])
        assert d["rank_ms_per_step_min"] <= d["rank_ms_per_step_max"] and d["local_step_ms"] > 0 and abs(d["allreduce_exposed_ms"]) < 0.5 * d["local_step_ms"]


def test_bench_gpus_n_spawns_n_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (round-3 verdict: the flag used to be parsed
    and ignored).  --spawn-probe takes the same re-exec path but joins a gloo group and counts the ranks instead of touching a GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-probe"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out == {"probe": True, "n_gpus": 2, "rccl_ranks": 2, "spawned": True}


def test_bench_gpus_n_c_backend_probe_and_rank_count_check():
    """Round-5 verdict item 9: the C-ABI transport's launch path (`--dist-backend c`: gloo rendezvous, rank 0's unique id shipped with broadcast_object_list) through
    the same re-exec as the torch backend, and the rank-count check of the multi-rank line -- a launcher that started fewer ranks than --gpus promises is a
    non-zero exit with a message, not a line of record."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "c", "--spawn-probe"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out == {"probe": True, "n_gpus": 2, "rccl_ranks": 2, "spawned": True, "dist_backend": "c", "uid_broadcast_ok": True}
    # two ranks started by a launcher, --gpus 4 on the command line: the probe (like the real line) must exit non-zero
    from bench import _free_port
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dist-backend", "c", "--spawn-probe"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "not a 4-GPU measurement" in (r.stderr + r.stdout), (r.returncode, r.stderr[-1500:])


def test_bench_gpus_n_refuses_without_devices():
    """Without N visible GPUs the N-rank launch must refuse loudly instead of printing a one-rank number (no GPU in the dev container)."""
    import subprocess
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the refusal path does not apply")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
