import torch, ctypes as C, os, sys, time
sys.path.insert(0, '/root/repo')
from deeptreeattention_amd import Hang2020 as H, _lib
from deeptreeattention_amd.engine import FusedTrainer
m = H.Hang2020(369, 200, precision="bf16").cuda().train()
tr = FusedTrainer(m, lr=1e-4)
x = torch.rand(1024, 369, 11, 11, device="cuda"); y = torch.randint(0, 200, (1024,), device="cuda")
for _ in range(5): tr.train_step(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): tr.train_step(x, y)
torch.cuda.synchronize(); print(os.environ.get("DTA_PACK_CG"), "ms/step", (time.perf_counter() - t0) / 30 * 1e3)
