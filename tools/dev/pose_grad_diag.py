"""Per-tensor gradient error of a Pose train step against the oracle (fp32), worst first."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
from oracle import yolo_oracle as O
from yolosharp_amd import Engine, model as M
from test_obb_pose import make_ref, _load

family, size, B, H, W = int(sys.argv[1]), sys.argv[2], 2, int(sys.argv[3]), int(sys.argv[3])
eng = Engine()
name = f"Yolov{family}Pose"
NC = int(sys.argv[4]) if len(sys.argv) > 4 else 1
ref = make_ref(getattr(O, name), NC, size)
m = _load(eng, ref, getattr(M, name), NC, size, B, H, W)
x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(3))
batch = O.synthetic_batch(B, H, W, NC, seed=1, kmax=12)
batch["keypoints"] = O.synthetic_keypoints(batch)
nb = {k: v.numpy() for k, v in batch.items()}
m.train(); ref.train()
m.forward(x.numpy(), fetch=False)
_, rp = ref(x)
rp["kpts"].retain_grad(); rp["boxes"].retain_grad(); rp["scores"].retain_grad()
loss, items = M.v8PoseLoss(m)(None, nb)
rloss, ritems = O.v8PoseLoss(NC)(rp, batch)
print("items", items, ritems.numpy())
rloss.sum().backward()
for k in ("kpts", "boxes", "scores"):
    r = rp[k].grad.numpy(); g = m.get_output("d" + k)
    print(k, np.abs(g - r).max() / np.abs(r).max())
m.zero_grad(); m.backward()
grads = m.grads()
rows = []
for n, p in ref.named_parameters():
    if p.grad is None:
        continue
    r = p.grad.numpy()
    rows.append((np.abs(grads[n] - r).max() / max(np.abs(r).max(), 1e-30), n, float(np.abs(r).max())))
rows.sort(reverse=True)
for e, n, mx in rows[:14]:
    print(f"{e:.2e} {mx:.3e} {n}")
