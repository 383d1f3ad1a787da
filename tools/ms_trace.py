"""Developer tool: only the batched multi-stage step, for kernel traces.  python tools/ms_trace.py [B] [steps] [serial]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeptreeattention_amd  # noqa: E402
from deeptreeattention_amd.engine import MultiStageTrainer  # noqa: E402
from deeptreeattention_amd.year import learned_ensemble  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
serial = len(sys.argv) > 3
deeptreeattention_amd.set_default_precision("bf16")
dev = torch.device("cuda:0")
cfg = {"pretrain_state_dict": None, "bands": 369}
classes = [2, 2, 12, 7, 5]
tr = MultiStageTrainer([learned_ensemble(3, c, cfg).to(dev).train() for c in classes], [1e-6, 1e-6, 5e-6, 1e-4, 5e-6])
batch = [(None, {"HSI": [torch.rand(B, 369, 11, 11, device=dev) for _ in range(3)]}, torch.randint(0, c, (B,), device=dev))
         for c in classes]
present = [[True] * 3] * 5
for _ in range(steps):
    if serial:
        [tr.training_step(batch, 0, l, present[l]) for l in range(5)]
    else:
        tr.training_step_all(batch, 0, present)
torch.cuda.synchronize()
