"""Developer tool: phase cycle stamps of one conv2-forward workgroup (library built with -DDTA_TICKS)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeptreeattention_amd import Hang2020 as H, _lib
from deeptreeattention_amd.engine import FusedTrainer
m = H.Hang2020(369, 200, precision="bf16").cuda().train()
tr = FusedTrainer(m, lr=1e-4)
x = torch.rand(1024, 369, 11, 11, device="cuda"); y = torch.randint(0, 200, (1024,), device="cuda")
for _ in range(5): tr.train_step(x, y)
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_longlong * 16)()
L.dta_debug_cticks(buf)
t = [buf[i] for i in range(6)]
names = ["entry -> staging plan ready (kernel arguments, divisions, bias request)", "first chunk fetched + row tables + stored, barrier", "chunk loop", "output store", "statistics"]
for i, n in enumerate(names): print(f"{n:<76} {t[i + 1] - t[i]:>8} cycles")
print("total", t[5] - t[0])
print("epilogue split: loop end -> shift/bias ready", buf[6] - t[3], "| pack + statistics sums + LDS writes", buf[7] - buf[6], "| barrier", buf[8] - buf[7], "| LDS -> global stores issued", t[4] - buf[8])
