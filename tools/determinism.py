import torch, sys
sys.path.insert(0, '.')
from deeptreeattention_amd import Hang2020 as H
from deeptreeattention_amd.engine import FusedTrainer
for prec in ("bf16", "fp32"):
    torch.manual_seed(0)
    m = H.Hang2020(369, 200, precision=prec).cuda().train()
    tr = FusedTrainer(m, lr=1e-4, keep_grads=True)
    x = torch.rand(1024, 369, 11, 11, device="cuda"); y = torch.randint(0, 200, (1024,), device="cuda")
    res = []
    for rep in range(3):
        logits = tr._forward_scores(x); loss = tr._loss(logits, tr._labels(y), True); tr._backward(tr.dlogits)
        torch.cuda.synchronize()
        g = torch.cat([tr.g_head.clone(), tr.g_tail.clone()]) if tr.flat_g is None else tr.flat_g.clone()
        res.append((float(loss), g, float(tr.alpha_g)))
    for r in res[1:]:
        same = torch.equal(res[0][1], r[1])
        nd = int((res[0][1] != r[1]).sum())
        print(prec, "loss equal", res[0][0] == r[0], "grads bit-identical", same, "differing elements", nd, "of", r[1].numel(), "alpha_g", res[0][2], r[2])
