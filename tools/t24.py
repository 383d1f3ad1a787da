import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from oracle import hang2020_np as O, prng
from conftest import rel_l2
from test_hip_parity import make, grads_of, dev
g = np.load('/root/repo/tests/golden/subnets.npz')
for prec in ("fp32", "bf16"):
    bands, classes, B, hw = 16, 7, 2, 24
    m, p = make("spectral", bands, classes, 51, precision=prec)
    x = torch.from_numpy(prng.uniform01(52, hw, (B, bands, hw, hw))).to(dev())
    m.train()
    try:
        s = m(x)
        for i in range(3): print(prec, "head", i+1, rel_l2(s[i].detach().cpu().numpy(), g[f"spectral24/head{i+1}"]))
        ds = [torch.from_numpy(prng.uniform(52, 10 + i, (B, classes), -1, 1)).to(dev()) for i in range(3)]
        sum((a * b).sum() for a, b in zip(s, ds)).backward()
        worst = max((rel_l2(prm.grad.cpu().numpy(), g[f"spectral24/g/{k}"]), k) for k, prm in m.named_parameters() if not k.endswith("conv_layer.bias"))
        print(prec, "worst grad", worst)
    except Exception as e:
        print(prec, "ERROR", e)
