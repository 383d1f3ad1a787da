"""Developer tool: throughput of the year-ensemble train step (BASELINE configs[4] shape: 3 years of 369-band
crops), fused trainer vs the module-level path (autograd.Function per year + torch.optim.Adam)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeptreeattention_amd.engine import EnsembleTrainer  # noqa: E402
from deeptreeattention_amd.year import learned_ensemble  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 11
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
import deeptreeattention_amd  # noqa: E402
deeptreeattention_amd.set_default_precision(prec)
dev = torch.device("cuda:0")
cfg = {"pretrain_state_dict": None, "bands": 369}
imgs = [torch.rand(B, 369, HW, HW, device=dev) for _ in range(3)]
y = torch.randint(0, 200, (B,), device=dev)


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


m = learned_ensemble(3, 200, cfg).to(dev).train()
tr = EnsembleTrainer(m, lr=1e-4)
ms_sync = timed(lambda: tr.train_step(imgs, y), steps)
ms_nosync = timed(lambda: tr.train_step(imgs, y, present=[True, True, True]), steps)

m2 = learned_ensemble(3, 200, cfg).to(dev).train()
opt = torch.optim.Adam(m2.parameters(), lr=1e-4)


def module_step():
    opt.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(m2(imgs), y).backward()
    opt.step()


ms_mod = timed(module_step, steps)
import json  # noqa: E402
print(json.dumps({"workload": "year-ensemble train step, 3 x spectral_network(369, 200)", "per_gpu_batch": B, "crop": HW,
                  "dtype": prec, "steps": steps, "fused_ms_per_step_device_decided": round(ms_sync, 4),
                  "fused_ms_per_step_present_flags": round(ms_nosync, 4),
                  "crops_per_s_present_flags": round(B / ms_nosync * 1e3, 1),
                  "module_level_torch_adam_ms_per_step": round(ms_mod, 4)}))
