"""Developer tool: phase cycle sums of the fused-input first conv's chunk loop (library built with -DDTA_TICKS; set
DTA_LIB=<path> to load it).  Waves 0 (stages first) and 4 (multiplies first) of workgroup 100."""
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
L.dta_debug_xticks(buf)
names = ["tile-out (LDS reads + tile stores)", "multiply (when first)", "staging: wait, convert, LDS writes, fetches", "multiply (when second)", "barrier wait", "whole chunk loop"]
for w, tag in ((0, "wave 0 (stages first)"), (1, "wave 4 (multiplies first)")):
    print(tag)
    for i, n in enumerate(names):
        print(f"  {n:<48} {buf[w * 8 + i]:>9} cycles  ({buf[w * 8 + i] / 24:.0f} per chunk)")
b2 = (C.c_longlong * 8)()
L.dta_debug_xticks2(b2)
for w, tag in ((0, "wave 0"), (1, "wave 4")):
    print(tag, "staging split: input wait + convert + LDS writes %d, weight LDS writes %d, fetch issue %d (cycles per chunk)" % tuple(b2[w * 4 + k] // 24 for k in range(3)))
print("wave 0: entry -> chunk loop %d cycles; loop end -> output stored %d; loop end -> statistics done (exit) %d" % (buf[6], buf[7], buf[15]))
