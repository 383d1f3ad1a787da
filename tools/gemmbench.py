"""Time dta_linear_forward / backward at a few shapes (back-to-back launches, torch events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeptreeattention_amd import _lib
L = _lib.lib()
st = _lib.current_stream_ptr()
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [(1024, 32, 200), (1024, 128, 200), (1024, 512, 200), (1024, 2048, 200), (4096, 512, 200)]
if len(sys.argv) > 3: shapes = [tuple(int(v) for v in sys.argv[1:4])]
for (B, F, N) in shapes:
    x = torch.randn(B, F, device="cuda"); w = torch.randn(N, F, device="cuda"); b = torch.randn(N, device="cuda")
    out = torch.empty(B, N, device="cuda"); dout = torch.randn(B, N, device="cuda")
    dx = torch.empty(B, F, device="cuda"); gw = torch.zeros(N, F, device="cuda"); gb = torch.zeros(N, device="cuda")
    tf = timeit(lambda: L.dta_linear_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), B, F, N, _lib.ptr(out), st))
    ref = x @ w.t() + b
    err = (out - ref).abs().max().item()
    tdx = timeit(lambda: L.dta_linear_backward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(dout), B, F, N, _lib.ptr(dx), None, None, st))
    tgw = timeit(lambda: L.dta_linear_backward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(dout), B, F, N, None, _lib.ptr(gw), _lib.ptr(gb), st))
    tt = timeit(lambda: torch.addmm(b, x, w.t()))
    print(f"B={B} F={F} N={N}: fwd {tf:.1f} us (torch {tt:.1f}) dx {tdx:.1f} us  gw {tgw:.1f} us  err {err:.2e}")
