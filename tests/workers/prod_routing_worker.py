"""Worker of tests/test_production_routing.py: runs in its OWN process so that the conv-routing gates are the production ones (the
library reads YS_GEMM_MIN_M / YS_WGEMM_MIN_M / YS_F8_MIN_CIN / YS_F8_MIN_TAPS once, at the first convolution plan of a process, and
tests/conftest.py lowers them for the rest of the suite).  YOLOv8n, bf16, one training step's forward + loss + backward on the engine
and on the rounding-matched oracle (tests/bf16_ref.py); writes everything the test compares into an .npz.

usage: prod_routing_worker.py <out.npz> <B> <H> <W> [emu] [f32]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
for k in ("YS_GEMM_MIN_M", "YS_WGEMM_MIN_M", "YS_F8_MIN_CIN", "YS_F8_MIN_TAPS"):
    assert k not in os.environ, "%s is set: this process would not test production routing" % k

import numpy as np
import torch

from oracle import yolo_oracle as O
import bf16_ref as R


def main():
    out, B, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    emu = "emu" in sys.argv[5:]
    with_f32 = "f32" in sys.argv[5:]          # also run the fp32 engine on the same step (tests/test_production_routing.py, headline batch)
    from yolosharp_amd import Engine
    from yolosharp_amd.model import Yolov8, v8DetectionLoss
    if emu:
        from yolosharp_amd import build
        eng = Engine(lib_path=build.build_emu())
    else:
        eng = Engine(0)
        assert eng.is_device_build
    nc = 80
    torch.manual_seed(11)
    ref = O.Yolov8(nc=nc, size="n")
    for mod in ref.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.normal_(0, 0.1)
            mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(12))
    batch = O.synthetic_batch(B, H, W, nc, seed=13)
    m = Yolov8(eng, nc=nc, size="n", height=H, width=W, max_batch=B, dtype="bf16")
    m.load_state_dict({k: v.detach().numpy() for k, v in ref.state_dict().items()})
    eng.kernel_profile(True)
    m.train()
    _, preds = m.forward(x.numpy())
    loss, items = v8DetectionLoss(m)(None, {k: v.numpy() for k, v in batch.items()})
    m.zero_grad(); m.backward()
    grads = m.grads()
    lp = out + ".launches.csv"
    eng.kernel_profile_dump(lp)
    eng.kernel_profile(False)
    labels = [l for l in open(lp).read().splitlines()[1:]]
    os.remove(lp)
    res = {"items": np.asarray(items, np.float32), "boxes": preds["boxes"], "scores": preds["scores"], "labels": np.array(labels)}
    # ---- oracle, twice: rounding-matched (bf16 storage points) and plain fp32
    ref.train()
    def plain():
        # the fp32 engine is judged against the oracle in DOUBLE at the headline batch: at B = 64 a per-channel sum runs over 6.5 million terms and the float
        # oracle's own summation error (model.0.bn.bias: 1.2 % of the tensor's maximum against the fp32 engine) is larger than the 2e-3 being asserted
        if with_f32:
            ref.double()
            return ref(x.double())
        return ref(x)
    for tag, fwd in (("r", lambda: R.forward_bf16(ref, x)), ("f", plain)):
        ref.zero_grad()
        _, rp = fwd()
        rloss, ritems = O.v8DetectionLoss(nc)(rp, batch)
        rloss.sum().backward()                # the recorded graph already carries the gradient roundings (bf16_ref._RoundSTE)
        res[tag + "_items"] = ritems.detach().numpy()
        res[tag + "_boxes"] = rp["boxes"].detach().numpy(); res[tag + "_scores"] = rp["scores"].detach().numpy()
        for name, p in ref.named_parameters():
            if p.grad is not None:
                res[tag + "_g_" + name] = p.grad.numpy().copy()
    ref.float()
    for name, g in grads.items():
        res["e_g_" + name] = g
    m.close()
    if with_f32:
        m32 = Yolov8(eng, nc=nc, size="n", height=H, width=W, max_batch=B, dtype="f32")
        m32.load_state_dict({k: v.detach().numpy() for k, v in ref.state_dict().items()})
        m32.train()
        m32.forward(x.numpy(), fetch=False)
        _, items32 = v8DetectionLoss(m32)(None, {k: v.numpy() for k, v in batch.items()})
        m32.zero_grad(); m32.backward()
        res["items32"] = np.asarray(items32, np.float32)
        for name, g in m32.grads().items():
            res["e32_g_" + name] = g
        m32.close()
    np.savez(out, **res)


if __name__ == "__main__":
    main()
