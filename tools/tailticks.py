"""Developer tool (library built with -DDTA_TICKS): wall-clock stamps (10 ns ticks) of the fused forward tail's phases,
workgroups 0 and 200 of the last launch."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeptreeattention_amd import Hang2020 as H, _lib
from deeptreeattention_amd.engine import FusedTrainer
m = H.Hang2020(369, 200, precision="bf16").cuda().train()
tr = FusedTrainer(m, lr=1e-4, loss_weight=torch.ones(200))
x = torch.rand(1024, 369, 11, 11, device="cuda"); y = torch.randint(0, 200, (1024,), device="cuda")
for _ in range(8): tr.train_step(x, y)
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_longlong * 32)()
L.dta_debug_tail_ticks(buf)
names = ["loads issued -> coef barrier", "stage (BN, pool, spatial attention)", "matvec1", "h", "matvec2", "gate/features", "head product", "scores", "blend+CE+loss"]
for w in range(2):
    t = [buf[w * 16 + i] for i in range(16)]
    seq = [t[i] for i in range(9)] + [t[15]]
    print("workgroup", (0, 200)[w], " ".join(f"{n}: {(seq[i + 1] - seq[i]) * 0.01:.2f}us" for i, n in enumerate(names)), f"| total {(t[15] - t[0]) * 0.01:.2f}us")
