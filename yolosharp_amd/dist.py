"""Data-parallel gradient synchronisation (new functionality: the reference is single-device).

One process per GPU; the global batch is sharded by image; BatchNorm statistics stay per device (the reference has
no SyncBN); the single exchange step is a SUM all-reduce of the flat fp32 gradient buffer before AdamW
(each shard back-propagates B_local*items_local, so the sum reproduces the single-device global-batch gradient).
The engine's backward is split into segments (head -> neck -> late backbone -> stem; the stem is cut small because the last
all-reduce has no backward left to hide behind); the all-reduce of a finished segment is
issued asynchronously on RCCL's stream while the next segment's kernels run (torch.distributed backend "nccl" is
RCCL on ROCm; xGMI is point-to-point so a few large buckets beat many small ones).

The same class runs on CPU tensors with the gloo backend (world_size-2 tests)."""
import torch
import torch.distributed as dist


class _DevArray:
    """Zero-copy view of an engine-owned HIP buffer for torch (CUDA array interface)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def device_view(ptr, n, device):
    return torch.as_tensor(_DevArray(ptr, n), device=device)


def host_view(ptr, n):
    """The same view for the test-only interpreter build, whose "device" memory is host memory (gloo tests on CPU)."""
    import ctypes
    return torch.frombuffer((ctypes.c_float * int(n)).from_address(int(ptr)), dtype=torch.float32)


class GradSync:
    def __init__(self, flat_grads: torch.Tensor, segment_ranges, group=None, compress=None):
        """flat_grads: 1-D fp32 tensor aliasing the gradient buffer; segment_ranges: [(offset, count)] per backward segment.
        compress = "bf16": the exchange moves bf16 (half the xGMI bytes: a ring all-reduce is per-link bound); every rank reduces the
        same rounded values in the same order, so the ranks' weights stay bit-identical, but the summed gradient carries bf16
        rounding (2^-9 relative per addend) -- off by default, the fp32 exchange reproduces the single-device gradient."""
        assert compress in (None, "bf16")
        self.flat = flat_grads
        self.segs = [self.flat.narrow(0, int(o), int(c)) for o, c in segment_ranges]
        self.group = group
        self.compress = compress
        self.stage = [torch.empty_like(s, dtype=torch.bfloat16) for s in self.segs] if compress else None
        self.pending = []

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def allreduce_segment(self, seg):
        if self.world == 1:
            return
        if self.compress:
            self.stage[seg].copy_(self.segs[seg])
            self.pending.append((dist.all_reduce(self.stage[seg], op=dist.ReduceOp.SUM, group=self.group, async_op=True), seg))
        else:
            self.pending.append((dist.all_reduce(self.segs[seg], op=dist.ReduceOp.SUM, group=self.group, async_op=True), None))

    def wait(self):
        for w, seg in self.pending:
            w.wait()
            if seg is not None:
                self.segs[seg].copy_(self.stage[seg])
        self.pending = []


def train_step_dp(model, crit, sync: GradSync, images_dev, batch, labels_dev, lrs):
    """forward -> loss -> segmented backward overlapped with bucketed all-reduce -> AdamW -> zero_grad."""
    model.forward_device(images_dev, batch)
    crit.forward_device(*labels_dev)
    for seg in range(model.num_segments()):
        model.backward_segment(seg)
        sync.allreduce_segment(seg)
    sync.wait()
    model.adamw_step(lrs)
    model.zero_grad()
