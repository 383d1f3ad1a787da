"""Developer probe: how much would HIP-graph replay of the fused train step save?  (Adam's bias corrections are baked
into the captured launch here, so this measures time only.)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeptreeattention_amd import Hang2020 as H  # noqa: E402
from deeptreeattention_amd.engine import FusedTrainer  # noqa: E402

for B in (32, 128, 256, 1024):
    m = H.Hang2020(369, 200, precision="bf16").cuda().train()
    tr = FusedTrainer(m, lr=1e-4)
    x = torch.rand(B, 369, 11, 11, device="cuda")
    y = torch.randint(0, 200, (B,), device="cuda")
    for _ in range(5):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50 * 1e3
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        tr.train_step(x, y)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        tr.train_step(x, y)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 50 * 1e3
    print(f"B={B}: eager {eager:.3f} ms, graph replay {graph:.3f} ms")
