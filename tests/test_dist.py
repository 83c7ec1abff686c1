"""Data-parallel path: world_size-2 gloo test of GradSync -- the SUM all-reduce of per-shard gradients equals
the gradient of the single-device global batch when BN statistics are per shard (the reference has no SyncBN), and the
bucketed/segmented path equals one flat all-reduce."""
import os
import socket
import sys

import numpy as np
import pytest
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


def test_gradsync_single_process_is_identity():
    from yolosharp_amd.dist import GradSync
    g = torch.arange(10, dtype=torch.float32)
    s = GradSync(g, [(0, 4), (4, 6)])
    s.allreduce_segment(0); s.allreduce_segment(1); s.wait()
    assert torch.equal(g, torch.arange(10, dtype=torch.float32))


def test_segment_ranges_cover_flat_buffer(emu_lib_path):
    """The engine's backward segments partition the flat gradient buffer (head, neck, backbone)."""
    from yolosharp_amd import Engine
    from yolosharp_amd.model import Yolov8
    eng = Engine(lib_path=emu_lib_path)
    m = Yolov8(eng, nc=80, size="n", height=64, width=64, max_batch=1, dtype="f32")
    ranges = [m.segment_grad_range(s) for s in range(m.num_segments())]
    _, n = m.grad_buffer()
    assert ranges[0][0] == 0 and sum(c for _, c in ranges) == n == m.num_params()
    for (o1, c1), (o2, _) in zip(ranges, ranges[1:]):
        assert o1 + c1 == o2
    m.close()


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
