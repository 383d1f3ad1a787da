"""bf16 step at the bench shape (369 bands, 200 classes, B=1024): HIP gradients vs the bf16-mode oracle accumulating in
float64 and in float32, the two oracles against each other, per-tensor rel-L2 and norm errors."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import hang2020_np as O, prng
from deeptreeattention_amd import Hang2020 as H
bands, classes, B = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (369, 200, 1024)))
dev = torch.device("cuda:0")
p = O.init_params(O.hang2020_spec(bands, classes), seed=3)
m = H.Hang2020(bands, classes, precision="bf16")
m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in p.items()})
m = m.to(dev).train()
x = prng.uniform01(40 + B, 1, (B, bands, 11, 11)); y = prng.randint(40 + B, 2, (B,), classes)
w = (0.1 + (np.arange(classes) % 7)).astype(np.float32)
logits = m(torch.from_numpy(x).to(dev))
loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(y).to(dev), weight=torch.from_numpy(w).to(dev))
loss.backward()
got = {k: (None if q.grad is None else q.grad.detach().cpu().numpy().astype(np.float64)) for k, q in m.named_parameters()}
def orc(dt):
    O.bf16_mode(True)
    try:
        l, c, _ = O.hang2020_fwd(p, x, True, dt)
        _, dl = O.weighted_cross_entropy(l, y, w)
        return {k: np.asarray(v, np.float64) for k, v in O.hang2020_bwd(p, c, dl, dt).items()}
    finally:
        O.bf16_mode(False)
g64, g32 = orc(np.float64), orc(np.float32)
def whole(a, b):
    num = den = 0.0
    for k, v in b.items():
        if k.endswith("conv_layer.bias") or not np.any(v) or a.get(k) is None: continue
        num += ((a[k] - v) ** 2).sum(); den += (v ** 2).sum()
    return np.sqrt(num / den)
print(f"whole: HIP vs o64 {whole(got, g64):.3e}  HIP vs o32 {whole(got, g32):.3e}  o32 vs o64 {whole(g32, g64):.3e}")
rows = []
for k, v in g64.items():
    if k.endswith("conv_layer.bias") or not np.any(v) or got.get(k) is None or v.size < 1000: continue
    n = np.linalg.norm(v)
    rows.append((k, v.size, np.linalg.norm(got[k] - v) / n, abs(np.linalg.norm(got[k]) - n) / n,
                 np.linalg.norm(got[k] - g32[k]) / np.linalg.norm(g32[k]), abs(np.linalg.norm(got[k]) - np.linalg.norm(g32[k])) / np.linalg.norm(g32[k]),
                 np.linalg.norm(g32[k] - v) / n, abs(np.linalg.norm(g32[k]) - n) / n))
print(f"{'tensor':58s} {'size':>8s}  rel(HIP,o64) norm(HIP,o64) rel(HIP,o32) norm(HIP,o32) rel(o32,o64) norm(o32,o64)")
for r in sorted(rows, key=lambda r: -r[2]):
    print(f"{r[0]:58s} {r[1]:8d}  {r[2]:.2e}     {r[3]:.2e}      {r[4]:.2e}     {r[5]:.2e}      {r[6]:.2e}     {r[7]:.2e}")
