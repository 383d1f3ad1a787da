"""Developer tool: the module-level step (Hang2020 module + optim.cross_entropy + DtaAdam) at B = 1024 bf16: wall clock per step
over 300 steps, host-side enqueue time per step (no synchronisation), and the same for the fused trainer."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeptreeattention_amd import Hang2020 as H
from deeptreeattention_amd.engine import FusedTrainer
from deeptreeattention_amd.optim import DtaAdam, cross_entropy
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(77)
x = torch.rand(1024, 369, 11, 11, device=dev, generator=g); y = torch.randint(0, 200, (1024,), device=dev, generator=g)
w = torch.ones(200, device=dev)


def timed(step, n=300):
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return round((t2 - t0) / n * 1e3, 4), round((t1 - t0) / n * 1e3, 4)


torch.manual_seed(1234)
m = H.Hang2020(369, 200, precision="bf16").to(dev).train()
opt = DtaAdam(m.parameters(), lr=1e-4)


def step_dta():
    opt.zero_grad()
    loss = cross_entropy(m(x), y, weight=w)
    loss.backward()
    opt.step()


print("module path (DtaAdam + optim.cross_entropy): ms/step, host enqueue ms/step:", timed(step_dta))
opt.close()
torch.manual_seed(1234)
m2 = H.Hang2020(369, 200, precision="bf16").to(dev).train()
tr = FusedTrainer(m2, lr=1e-4, loss_weight=w)
print("fused trainer:                               ms/step, host enqueue ms/step:", timed(lambda: tr.train_step(x, y)))
