import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from deeptreeattention_amd import Hang2020 as H
from deeptreeattention_amd.engine import FusedTrainer
def run(seed, steps=4, B=530, classes=12, bands=40):
    torch.manual_seed(7)
    m = H.Hang2020(bands, classes, precision="bf16").cuda().train()
    tr = FusedTrainer(m, lr=1e-3, loss_weight=torch.ones(classes).cuda())
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    x = torch.rand(B, bands, 11, 11, device="cuda", generator=g)
    y = torch.randint(0, classes, (B,), device="cuda", generator=g)
    for _ in range(steps):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    return {k: v.detach().clone().cpu() for k, v in m.state_dict().items()}
a = run(0); b = run(0); c = run(0)
bad = [k for k in a if not (torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]))]
print("NWT", os.environ.get("DTA_TAIL_NWT"), "nondeterministic tensors:", len(bad), bad[:6])
