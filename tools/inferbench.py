"""Developer tool: inference throughput (eval-mode forward + softmax/top-2 on device), the reference's predict_step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeptreeattention_amd import Hang2020 as H  # noqa: E402
from deeptreeattention_amd.engine import predict, Predictor  # noqa: E402

dev = torch.device("cuda:0")
for prec in ("bf16", "fp32"):
    m = H.Hang2020(369, 200, precision=prec).to(dev).eval()
    for B in (128, 1024, 4096):
        x = torch.rand(B, 369, 11, 11, device=dev)
        for _ in range(3):
            predict(m, x, return_probs=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            predict(m, x, return_probs=False)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        pr = Predictor(m)
        for _ in range(3):
            pr(x, return_probs=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            pr(x, return_probs=False)
        torch.cuda.synchronize()
        ms2 = (time.perf_counter() - t0) / n * 1e3
        fz = Predictor(m, frozen=True)       # weight re-layouts kept in the workspace (DTA_REUSE_PACKED)
        for _ in range(3):
            fz(x, return_probs=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fz(x, return_probs=False)
        torch.cuda.synchronize()
        ms3 = (time.perf_counter() - t0) / n * 1e3
        print(f"{prec} B={B}: predict() {ms:.3f} ms {B / ms * 1e3:,.0f} patches/s | Predictor {ms2:.3f} ms {B / ms2 * 1e3:,.0f} patches/s"
              f" | frozen weights {ms3:.3f} ms {B / ms3 * 1e3:,.0f} patches/s")
