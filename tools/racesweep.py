"""Developer tool: run-to-run deviation sweep over model kinds / batch sizes / precisions.  For every parameter
gradient the deviations from the second run are collected; an outlier (max >> median) hints at a race."""
import sys

import torch

from deeptreeattention_amd import Hang2020 as H

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150


def sweep(name, make, shape, classes, mode, heads_sum=False):
    torch.manual_seed(7)
    m = make().to(dev)
    m.train(mode == "train")
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    x = torch.rand(*shape, device=dev, generator=g)
    y = torch.randint(0, classes, (shape[0],), device=dev, generator=g)

    def grads():
        m.zero_grad(set_to_none=True)
        out = m(x)
        if isinstance(out, (list, tuple)):
            loss = sum(torch.nn.functional.cross_entropy(o, y) for o in out)
            out = out[-1]
        else:
            loss = torch.nn.functional.cross_entropy(out, y)
        loss.backward()
        return {k: p.grad.detach().double().clone() for k, p in m.named_parameters() if p.grad is not None}, out.detach().double()

    grads()
    ref, oref = grads()
    devs = {k: [] for k in ref}
    odev = 0.0
    for _ in range(iters):
        got, o = grads()
        odev = max(odev, float((o - oref).norm() / oref.norm()))
        for k in ref:
            n = float(ref[k].norm())
            if n > 0 and not k.endswith("conv_layer.bias"):
                devs[k].append(float((got[k] - ref[k]).norm()) / n)
    flagged = []
    for k, v in devs.items():
        if not v:
            continue
        t = torch.tensor(v)
        med, mx = float(t.median()), float(t.max())
        if mx > 1e-5 and mx > 8 * max(med, 1e-7):
            flagged.append((k, med, mx))
    worst = max((max(v) for v in devs.values() if v), default=0.0)
    print("%-44s out-dev %.1e  worst %.1e  %s" % (name, odev, worst, "OUTLIERS " + str(flagged) if flagged else "ok"), flush=True)


for prec in ("fp32", "bf16"):
    for mode in ("eval", "train"):
        for B in (37, 128, 512, 1024):
            sweep(f"hang2020 B={B} {prec} {mode}", lambda: H.Hang2020(369, 200, precision=prec), (B, 369, 11, 11), 200, mode)
        sweep(f"spectral24 B=16 {prec} {mode}", lambda: H.spectral_network(64, 20, precision=prec), (16, 64, 24, 24), 20, mode)
        sweep(f"spectral11 B=700 {prec} {mode}", lambda: H.spectral_network(369, 200, precision=prec), (700, 369, 11, 11), 200, mode)
        sweep(f"spatial11 B=700 {prec} {mode}", lambda: H.spatial_network(369, 200, precision=prec), (700, 369, 11, 11), 200, mode)
        sweep(f"vanilla B=600 {prec} {mode}", lambda: H.vanilla_CNN(5, 3, precision=prec), (600, 5, 11, 11), 3, mode)
