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
buf = (C.c_longlong * 64)()
L.dta_debug_wticks(buf)
for w in range(8):
    t = [buf[w * 8 + i] for i in range(6)]
    print("wave", w, "stageA", t[0], "ksteps", t[1], "stageB", t[2], "barrier", t[5], "loop total", t[3], "niter", t[4])
print("wave 0: entry -> loop", buf[6], "cycles; loop end -> k-slice hand-over done", buf[7], "; -> slab stores issued", buf[14], "; -> performed", buf[15])
