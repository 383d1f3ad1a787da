"""Time several kernel sites (HIP events inside the library) over train steps; prints avg us per site."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeptreeattention_amd import Hang2020 as H, _lib
from deeptreeattention_amd.engine import FusedTrainer
sites = [int(s) for s in sys.argv[1:]] or list(range(18))
m = H.Hang2020(369, 200, precision=os.environ.get("PREC", "bf16")).cuda().train()
tr = FusedTrainer(m, lr=1e-4)
x = torch.rand(1024, 369, 11, 11, device="cuda"); y = torch.randint(0, 200, (1024,), device="cuda")
for _ in range(5): tr.train_step(x, y)
torch.cuda.synchronize()
L = _lib.lib()
for site in sites:
    L.dta_profile_enable(-1); L.dta_profile_enable(site)
    for _ in range(20): tr.train_step(x, y)
    torch.cuda.synchronize()
    buf = (C.c_float * 512)(); n = L.dta_profile_collect(buf, 512)
    if n: print("site", site, "avg us %.1f" % (1e3 * sum(buf[i] for i in range(n)) / n), "n", n)
L.dta_profile_enable(-1)
