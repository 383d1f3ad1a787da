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
L.dta_debug_ticks(buf)
# the last stage_bwd launch is layer 0
for g in range(2):
    t = [buf[g * 32 + i] for i in range(9)]
    print("group", g, "fwd", t[1]-t[0], "loadD", t[2]-t[1], "att_bwd", t[3]-t[2], "final", t[4]-t[3], "bnpart", t[5]-t[4], "total", t[5]-t[0],
          "| spatial: ds", t[6]-t[2], "stencils", t[7]-t[6], "vec", t[8]-t[7], "Dupd", t[3]-t[8])
