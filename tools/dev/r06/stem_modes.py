import sys, os, time
for k,v in dict(YS_GEMM_MIN_M="1",YS_HALO_MIN_FILL="1",YS_WGEMM_MIN_M="1",YS_F8_MIN_CIN="32",YS_F8_MIN_TAPS="1").items(): os.environ.setdefault(k,v)
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from oracle import yolo_oracle as O
from yolosharp_amd import Engine, build as BB
from test_model import make_ref, build, relerr
from yolosharp_amd.model import v8DetectionLoss
engine = Engine(0)
def run(B,H,W,damp):
    nc=80
    ref = make_ref(seed=3)
    if damp:
        for mod in ref.modules():
            if isinstance(mod, O.Bottleneck): mod.cv2.bn.weight.data.mul_(damp)
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(4))
    batch = O.synthetic_batch(B, H, W, nc, seed=5, kmax=5)
    G=[]; t=time.time()
    for mode in (1,0):
        with engine.options(STEM_DIRECT=mode):
            m = build(engine, ref, H, W, B, "bf16")
        m.train(); m.forward(x.numpy())
        loss, items = v8DetectionLoss(m)(None, {k: v.numpy() for k, v in batch.items()})
        m.zero_grad(); m.backward(); g=m.grads(); G.append((g["model.0.conv.weight"].copy(), g["model.0.bn.weight"].copy(), items)); m.close()
    a,b=G
    print(B,H,W,damp,"relerr gw %.4f gg %.4f cos %.5f items"%(relerr(a[0],b[0]), relerr(a[1],b[1]), float((a[0]*b[0]).sum()/np.sqrt((a[0]**2).sum()*(b[0]**2).sum()))), a[2], b[2], "%.0fs"%(time.time()-t), flush=True)
for cfg in [(2,96,160,0),(2,96,160,0.25),(4,128,160,0),(4,128,160,0.25),(8,160,160,0),(8,160,160,0.25),(8,224,288,0),(8,224,288,0.25),(16,224,288,0.25)]: run(*cfg)
