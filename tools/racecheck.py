"""Developer tool: run the same backward many times and report per-parameter run-to-run deviation.  Split-K atomics
reorder fp32 sums (~1e-6 relative); anything near 1/B hints at a patch-level race."""
import sys

import torch

from deeptreeattention_amd import Hang2020 as H

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
mode = sys.argv[4] if len(sys.argv) > 4 else "eval"
dev = torch.device("cuda:0")
torch.manual_seed(7)
m = H.Hang2020(369, 200, precision=prec).to(dev)
m.train(mode == "train")
g = torch.Generator(device=dev)
g.manual_seed(11)
x = torch.rand(B, 369, 11, 11, device=dev, generator=g)
y = torch.randint(0, 200, (B,), device=dev, generator=g)


import ctypes as C
from deeptreeattention_amd import _lib
poison = len(sys.argv) > 5 and sys.argv[5] == "poison"
desc = _lib.NetDesc(B, 369, 11, 11, 200, _lib.NET_HANG2020, _lib.dtype_code(prec), 1 if mode == "train" else 0, 4,
                    H.BN_MOMENTUM, H.BN_EPS)
ws_bytes = _lib.lib().dta_net_workspace_bytes(C.byref(desc))


def grads():
    if poison:   # the next workspace allocation reuses this block: any read of bytes the kernels did not write is a NaN
        junk = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        junk.fill_(0xFF)
        del junk
    m.zero_grad(set_to_none=True)
    out = m(x)
    torch.nn.functional.cross_entropy(out, y).backward()
    return {k: p.grad.detach().double().clone() for k, p in m.named_parameters() if p.grad is not None}, out.detach().double()


ref, oref = grads()
if poison:
    bad = [k for k, v in ref.items() if not torch.isfinite(v).all()]
    print("poisoned workspace: non-finite logits", bool(not torch.isfinite(oref).all()), "non-finite grads", bad)
second, _ = grads()
for k in ref:
    n = float(ref[k].norm())
    if n > 0 and float((second[k] - ref[k]).norm()) / n > 2e-5:
        print("run0 vs run1", k, "%.3e" % (float((second[k] - ref[k]).norm()) / n))
ref = second
worst = {}
for it in range(iters):
    got, o = grads()
    e = float((o - oref).norm() / oref.norm())
    if e > 1e-6:
        print("iter", it, "logits dev", e)
    for k in ref:
        n = float(ref[k].norm())
        if n == 0:
            continue
        e = float((got[k] - ref[k]).norm()) / n
        if e > 2e-5:
            print("iter", it, k, "%.3e" % e)
        worst[k] = max(worst.get(k, 0.0), e)
top = sorted(worst.items(), key=lambda kv: -kv[1])[:8]
for k, e in top:
    print("%-60s %.3e" % (k, e))
