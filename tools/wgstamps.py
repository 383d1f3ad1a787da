"""Developer tool: when does every workgroup of a kernel start and end?  (library built with -DDTA_TICKS)
Prints, per instrumented kernel of the LAST train step: workgroups, kernel span, start-time spread, duration
distribution and how many workgroups were running at a few instants."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeptreeattention_amd import Hang2020 as H, _lib
from deeptreeattention_amd.engine import FusedTrainer
m = H.Hang2020(369, 200, precision="bf16").cuda().train()
tr = FusedTrainer(m, lr=1e-4)
x = torch.rand(1024, 369, 11, 11, device="cuda"); y = torch.randint(0, 200, (1024,), device="cuda")
for _ in range(5): tr.train_step(x, y)
torch.cuda.synchronize()
L = _lib.lib()
names = {"conv": ["conv1 fwd <2,2,fused input>", "conv2 fwd", "conv3 fwd", "conv1 wgrad"],
         "stage": ["stage bwd lean C=32", "stage bwd lean C=64", "stage fwd lean C=32", "bn apply (first stage)"]}
for tu in ("conv", "stage"):
    buf = np.zeros((4, 8192, 2), dtype=np.int64)
    getattr(L, "dta_debug_wgstamps_" + tu)(buf.ctypes.data_as(C.c_void_p))
    for i, n in enumerate(names[tu]):
        s = buf[i]; s = s[s[:, 1] > 0]
        if not len(s): continue
        t0 = s[:, 0].min(); st = (s[:, 0] - t0) / 100.0; en = (s[:, 1] - t0) / 100.0; du = en - st     # us
        q = lambda a, p: float(np.percentile(a, p))
        print(f"{n}: {len(s)} workgroups, span {en.max():.1f} us")
        print(f"   start  p50 {q(st,50):6.1f}  p90 {q(st,90):6.1f}  max {st.max():6.1f} us")
        print(f"   length min {du.min():6.1f}  p50 {q(du,50):6.1f}  p90 {q(du,90):6.1f}  max {du.max():6.1f} us")
        ts = np.linspace(0, en.max(), 9)[1:-1]
        print("   running at " + "  ".join(f"{t:.0f}us:{int(((st <= t) & (en > t)).sum())}" for t in ts))
