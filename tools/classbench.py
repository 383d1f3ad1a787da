"""Developer tool: FusedTrainer step time (bf16, B=1024, 369 bands) over class counts, aligned (multiple of 4) or not."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeptreeattention_amd import Hang2020 as H
from deeptreeattention_amd.engine import FusedTrainer
for classes in [int(c) for c in (sys.argv[1:] or ["200", "199", "43", "44"])]:
    torch.manual_seed(0)
    m = H.Hang2020(369, classes, precision="bf16").cuda().train()
    tr = FusedTrainer(m, lr=1e-4)
    x = torch.rand(1024, 369, 11, 11, device="cuda"); y = torch.randint(0, classes, (1024,), device="cuda")
    for _ in range(30): tr.train_step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): tr.train_step(x, y)
    torch.cuda.synchronize(); print("classes", classes, "ms/step %.4f" % ((time.perf_counter() - t0) / 200 * 1e3))
    del tr, m
