"""Developer probe: what HBM bandwidth do plain streaming kernels reach on this part?  (torch copy = read + write,
torch sum = read only, fill = write only; sizes from 32 MB -- a kernel the size of ours -- to 2 GB)"""
import torch
dev = "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (32, 128, 300, 1024, 2048):
    n = mb * 1024 * 1024 // 4
    a = torch.rand(n, device=dev); b = torch.empty_like(a)
    tc = t(lambda: b.copy_(a)); ts = t(lambda: a.sum()); tf = t(lambda: b.fill_(1.0))
    print(f"{mb:5d} MB: copy {2 * mb / 1024 / tc / 1e3 * 1.048576:.2f} TB/s ({tc * 1e6:.1f} us)  read(sum) {mb / 1024 / ts / 1e3 * 1.048576:.2f} TB/s  write(fill) {mb / 1024 / tf / 1e3 * 1.048576:.2f} TB/s")
