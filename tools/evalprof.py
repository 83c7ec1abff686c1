import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); os.chdir(sys.path[0])
from yolosharp_amd import Engine
from yolosharp_amd.model import Yolov8
eng = Engine(0)
B=64
m = Yolov8(eng, nc=80, size="n", height=640, width=640, max_batch=B, dtype="bf16")
m.init_weights(2); m.eval()
x = np.random.default_rng(0).random((B,3,640,640), dtype=np.float32)
d = eng.to_device(x)
for _ in range(3): m.forward_device(d, B)
eng.synchronize()
eng.kernel_profile(True)
for _ in range(2): m.forward_device(d, B)
eng.synchronize()
print(eng.kernel_profile_read("conv_igemm"))
eng.kernel_profile_dump("gpurun_out/eval_launches.csv")
eng.kernel_profile(False)
