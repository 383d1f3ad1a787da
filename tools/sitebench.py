"""Time one kernel site (HIP events inside the library) over N train steps; prints avg ms."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeptreeattention_amd import Hang2020 as H, _lib
from deeptreeattention_amd.engine import FusedTrainer
site = int(sys.argv[1]) if len(sys.argv) > 1 else 0
m = H.Hang2020(369, 200, precision="bf16").cuda().train()
tr = FusedTrainer(m, lr=1e-4)
x = torch.rand(1024, 369, 11, 11, device="cuda"); y = torch.randint(0, 200, (1024,), device="cuda")
for _ in range(5): tr.train_step(x, y)
torch.cuda.synchronize()
L = _lib.lib(); L.dta_profile_enable(-1); L.dta_profile_enable(site)
for _ in range(30): tr.train_step(x, y)
torch.cuda.synchronize()
buf = (C.c_float * 512)(); n = L.dta_profile_collect(buf, 512)
print("ABLATE", os.environ.get("DTA_ABLATE"), "site", site, "avg us", 1e3 * sum(buf[i] for i in range(n)) / n)
