"""Developer tool: MultiStage.predict_step (5 levels x 3 years, the same crops for every level) as one launch chain vs one
Predictor per level.  python tools/ms_predict_bench.py [B] [steps]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeptreeattention_amd  # noqa: E402
from deeptreeattention_amd.engine import MultiStagePredictor, Predictor  # noqa: E402
from deeptreeattention_amd.year import learned_ensemble  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
deeptreeattention_amd.set_default_precision("bf16")
dev = torch.device("cuda:0")
cfg = {"pretrain_state_dict": None, "bands": 369}
models = [learned_ensemble(3, c, cfg).to(dev).eval() for c in [2, 2, 12, 7, 5]]
x = [torch.rand(B, 369, 11, 11, device=dev) for _ in range(3)]
per = [Predictor(m) for m in models]
one = MultiStagePredictor(models)
frz = MultiStagePredictor(models, frozen=True)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


a = timed(lambda: [p(x) for p in per]); b = timed(lambda: one(x)); c = timed(lambda: frz(x))
print(json.dumps({"workload": "MultiStage.predict_step: 5 levels x 3 years, 369 bands, 11x11, bf16", "batch": B, "per_level_ms": round(a, 4),
                  "one_chain_ms": round(b, 4), "speedup": round(a / b, 2), "crops_per_s_one_chain": round(B / b * 1e3, 1),
                  "one_chain_frozen_weights_ms": round(c, 4), "crops_per_s_frozen": round(B / c * 1e3, 1)}))
