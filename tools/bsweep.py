"""Developer tool: fused train step time at several batch sizes (bf16)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeptreeattention_amd import Hang2020 as H  # noqa: E402
from deeptreeattention_amd.engine import FusedTrainer  # noqa: E402

for B in [int(v) for v in sys.argv[1:]] or [128, 256, 512, 1024]:
    m = H.Hang2020(369, 200, precision="bf16").cuda().train()
    tr = FusedTrainer(m, lr=1e-4)
    x = torch.rand(B, 369, 11, 11, device="cuda")
    y = torch.randint(0, 200, (B,), device="cuda")
    for _ in range(5):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    print(f"B={B}: {(time.perf_counter() - t0) / 40 * 1e3:.4f} ms")
