"""Developer tool: which parameters differ bitwise after ONE step of the fused trainer vs the module path (DtaAdam)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeptreeattention_amd import Hang2020 as H
from deeptreeattention_amd.engine import FusedTrainer
from deeptreeattention_amd.optim import DtaAdam, cross_entropy
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
bands, classes, B, lr = 20, 7, 6, 1e-3
torch.manual_seed(5)
m1 = H.Hang2020(bands, classes, precision=prec).cuda().train()
m2 = H.Hang2020(bands, classes, precision=prec).cuda().train()
m2.load_state_dict(m1.state_dict())
w = torch.linspace(0.1, 1.0, classes).cuda()
tr = FusedTrainer(m1, lr=lr, loss_weight=w)
opt = DtaAdam(m2.parameters(), lr=lr, fuse_zero_grad=False)
x = torch.rand(B, bands, 11, 11, device="cuda"); y = torch.randint(0, classes, (B,), device="cuda")
for step in range(2):
    l1 = tr.train_step(x, y)
    opt.zero_grad()
    out = m2(x)
    out.retain_grad()
    l2 = cross_entropy(out, y, weight=w); l2.backward()
    torch.cuda.synchronize()
    print("  joint scores equal:", torch.equal(out.detach(), tr.logits), " dlogits equal:", torch.equal(out.grad, tr.dlogits),
          " max|d dlogits|", float((out.grad - tr.dlogits).abs().max()))
    g2 = {k: p.grad.detach().clone() for k, p in m2.named_parameters()}
    opt.step()
    torch.cuda.synchronize()
    print("step", step, "loss", float(l1), float(l2), "equal", float(l1) == float(l2))
    g1 = tr.grads_by_name() if hasattr(tr, "grads_by_name") else None
    bad = [k for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()) if not torch.equal(a, b)]
    print("  params differing bitwise:", len(bad), bad[:8])
