"""Developer tool: the reference's multi-stage step (5 levels x 3 years of spectral_network(369, classes_l), 11x11 crops;
train.py:75-100) as ONE launch chain (MultiStageTrainer.training_step_all) against the level-by-level step.
    python tools/multistagebench.py [B] [steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeptreeattention_amd  # noqa: E402
from deeptreeattention_amd.engine import MultiStageTrainer  # noqa: E402
from deeptreeattention_amd.year import learned_ensemble  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
deeptreeattention_amd.set_default_precision("bf16")
dev = torch.device("cuda:0")
cfg = {"pretrain_state_dict": None, "bands": 369}
classes = [2, 2, 12, 7, 5]
tr = MultiStageTrainer([learned_ensemble(3, c, cfg).to(dev).train() for c in classes], [1e-6, 1e-6, 5e-6, 1e-4, 5e-6])
batch = [(None, {"HSI": [torch.rand(B, 369, 11, 11, device=dev) for _ in range(3)]}, torch.randint(0, c, (B,), device=dev))
         for c in classes]
present = [[True] * 3] * 5


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {"workload": "multi-stage step: 5 levels x 3 years x spectral_network(369, c), 11x11, bf16", "per_level_batch": B, "steps": steps}
res["serial_present_ms"] = round(timed(lambda: [tr.training_step(batch, 0, l, present[l]) for l in range(5)], steps), 4)
res["batched_present_ms"] = round(timed(lambda: tr.training_step_all(batch, 0, present), steps), 4)
res["serial_device_decided_ms"] = round(timed(lambda: [tr.training_step(batch, 0, l) for l in range(5)], steps), 4)
res["batched_device_decided_ms"] = round(timed(lambda: tr.training_step_all(batch, 0), steps), 4)
res["speedup_present"] = round(res["serial_present_ms"] / res["batched_present_ms"], 2)
res["crops_per_s_batched"] = round(5 * B / res["batched_present_ms"] * 1e3, 1)
print(json.dumps(res))
