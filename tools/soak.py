"""Developer tool: a few thousand steps of every trainer on one fixed batch -- loss must fall, nothing may go non-finite, device
memory must not grow (leaks in the per-step Python plumbing)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeptreeattention_amd import Hang2020 as H
from deeptreeattention_amd.engine import FusedTrainer, EnsembleTrainer, MetadataTrainer
from deeptreeattention_amd.metadata import metadata_sensor_fusion
from deeptreeattention_amd.optim import DtaAdam, cross_entropy
from deeptreeattention_amd.year import learned_ensemble
dev = torch.device("cuda", 0)
torch.manual_seed(3)
B, bands, classes = 256, 64, 24
x = torch.rand(B, bands, 11, 11, device=dev); y = torch.randint(0, classes, (B,), device=dev); site = torch.randint(0, 9, (B,), device=dev)

def run(name, step, n):
    first = last = None
    mem0 = None
    for i in range(n):
        l = step()
        if i == 20:
            torch.cuda.synchronize(); mem0 = torch.cuda.memory_allocated()
        if i % (n // 4) == 0 or i == n - 1:
            v = float(l)
            assert v == v and abs(v) < 1e6, (name, i, v)
            first = v if first is None else first
            last = v
    torch.cuda.synchronize()
    grew = torch.cuda.memory_allocated() - mem0
    print(f"{name:<28} {n} steps  loss {first:.4f} -> {last:.4f}  device memory growth after step 20: {grew} B")
    assert last < first and grew <= 1 << 20, name

m = H.Hang2020(bands, classes, precision="bf16").to(dev).train(); tr = FusedTrainer(m, lr=1e-3)
run("FusedTrainer bf16", lambda: tr.train_step(x, y), 2000)
m = H.Hang2020(bands, classes, precision="fp32").to(dev).train(); tr = FusedTrainer(m, lr=1e-3)
run("FusedTrainer fp32", lambda: tr.train_step(x, y), 600)
m = H.Hang2020(bands, classes, precision="bf16").to(dev).train(); opt = DtaAdam(m.parameters(), lr=1e-3)
def mod():
    opt.zero_grad(); l = cross_entropy(m(x), y); l.backward(); opt.step(); return l.detach()
run("module path + DtaAdam", mod, 2000)
m = metadata_sensor_fusion(bands=bands, sites=9, classes=classes, precision="bf16").to(dev).train(); tr = MetadataTrainer(m, lr=1e-3)
run("MetadataTrainer (native head)", lambda: tr.train_step(x, site, y), 2000)
m = learned_ensemble(3, classes, {"pretrain_state_dict": None, "bands": bands}).to(dev).train(); tr = EnsembleTrainer(m, lr=1e-3, loss_weight=torch.ones(classes))
xs = [x, torch.zeros_like(x), x * 0.5]
run("EnsembleTrainer (a zero year)", lambda: tr.train_step(xs, y), 1000)
print("soak ok")
