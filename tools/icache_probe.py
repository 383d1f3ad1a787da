"""Developer probe: what a launch of a 60 KB kernel costs when the instruction cache holds other kernels.  The head weight
gradient GEMM (k_gemm_group, 1024 x 512 x 200: 112 workgroups) timed with an event pair (a) right after a launch of itself,
(b) after a 1 GB device copy (data caches cold, code warm), (c) after a whole train step (26 other kernels: code and data cold),
(d) after a train step followed by an untimed launch of itself (code warm again, data mostly cold)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeptreeattention_amd import Hang2020 as H, _lib
from deeptreeattention_amd.engine import FusedTrainer
L = _lib.lib()
st = _lib.current_stream_ptr()
B, F, N = 1024, 512, 200
x = torch.randn(B, F, device="cuda"); w = torch.randn(N, F, device="cuda"); dout = torch.randn(B, N, device="cuda")
gw = torch.zeros(N, F, device="cuda"); gb = torch.zeros(N, device="cuda")
x2 = torch.randn(B, F, device="cuda"); dout2 = torch.randn(B, N, device="cuda")
gemm = lambda: L.dta_linear_backward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(dout), B, F, N, None, _lib.ptr(gw), _lib.ptr(gb), st)
gemm_other = lambda: L.dta_linear_backward(_lib.ptr(x2), _lib.ptr(w), _lib.ptr(dout2), B, F, N, None, _lib.ptr(gw), _lib.ptr(gb), st)
big = torch.empty(256 << 20, dtype=torch.float32, device="cuda"); big2 = torch.empty_like(big)
m = H.Hang2020(369, 200, precision="bf16").cuda().train()
tr = FusedTrainer(m, lr=1e-4, loss_weight=torch.ones(200))
xb = torch.rand(1024, 369, 11, 11, device="cuda"); yb = torch.randint(0, 200, (1024,), device="cuda")
def measure(before, n=60):
    ts = []
    for _ in range(n):
        before()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gemm(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
for _ in range(5): tr.train_step(xb, yb); gemm()
torch.cuda.synchronize()
def empty_pair(n=60):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
print(f"empty event pair {empty_pair():.1f} us")
print(f"(a) after itself (same operands)            {measure(gemm):.1f} us")
print(f"(a') after itself on other operands          {measure(gemm_other):.1f} us")
print(f"(b) after a 1 GB copy                        {measure(lambda: big2.copy_(big)):.1f} us")
print(f"(c) after a train step                       {measure(lambda: tr.train_step(xb, yb)):.1f} us")
print(f"(d) after a train step + itself (other data) {measure(lambda: (tr.train_step(xb, yb), gemm_other())):.1f} us")
print(f"(e) after a train step + 1 GB copy           {measure(lambda: (tr.train_step(xb, yb), big2.copy_(big))):.1f} us")
