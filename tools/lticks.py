"""Developer tool: phase cycle stamps of one lean stage-backward workgroup of the 11x11x32 stage, spectral and spatial
group (library built with -DDTA_TICKS)."""
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
buf = (C.c_longlong * 32)()
L.dta_debug_lticks(buf)
names = ["issue loads, stage weights/state", "first barrier (loads land)", "recompute BN/ReLU", "D*z sums (colsum | pixel sums)",
         "mat-vecs | stencils", "vec outputs + dwc / dK", "dv stores + LDS", "BN partial colsums"]
for g, kind in enumerate(("spectral", "spatial")):
    t = [buf[g * 16 + i] for i in range(9)]
    print(kind, "total", t[8] - t[0])
    for i, n in enumerate(names): print(f"   {n:<40} {t[i + 1] - t[i]:>8}")
