import os, sys
sys.path.insert(0, "/root/repo")
os.environ.setdefault("YS_GEMM_MIN_M", "1")
import numpy as np, torch
from yolosharp_amd import Engine
eng = Engine(0)
g = torch.Generator().manual_seed(0)
eng.kernel_profile(True)
x = torch.randn(1, 128, 12, 12, generator=g); w = torch.randn(160, 128, 3, 3, generator=g) * 0.1
y = eng.conv_bn_act(x.numpy(), w.numpy(), 3, 1, bn=None, bias=np.zeros(160, np.float32), act=False, dtype="bf16")
ref = torch.nn.functional.conv2d(x.bfloat16().float(), w.bfloat16().float(), padding=1).numpy()
print("err", np.abs(y - ref).max(), np.abs(ref).max())
eng.kernel_profile_dump("/tmp/l.csv"); print(open("/tmp/l.csv").read())
