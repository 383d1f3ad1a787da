"""Protocol time of the peer exchange launch (csrc/xchg.hip) for world = 1, 2, 4, 8 ranks SHARING one GPU: device-side
ticks of workgroup 0 from "all ranks present" to the end of the launch (dta_xchg_last_timing), i.e. without the launch
skew between the processes.  No xGMI here (peers are mapped on the same device): this is the protocol's latency chain
(flag round trips through uncached memory, two dependent pulls), not wire time."""
import os, sys, time, datetime, socket, ctypes as C
import numpy as np, torch
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = 900_788

def worker(rank, world, port, wgs, mode):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port); os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    torch.cuda.set_device(0)
    from deeptreeattention_amd.dist import PeerExchange
    from deeptreeattention_amd import _lib
    L = _lib.lib()
    ex = PeerExchange(N, timeout_s=10.0, max_workgroups=wgs)
    n = ex.capacity
    p = torch.randn(n, device="cuda"); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    ws, es = [], []
    for step in range(1, 41):
        ex.grad.fill_(float(rank + 1))
        torch.cuda.synchronize(); dist.barrier()
        if mode == "adam":
            ex.adam_step(p, m, v, None, None, -1, None, None, step, 1e-3, (0.9, 0.999), 1e-8, True)
        else:
            ex.allreduce()
        torch.cuda.synchronize()
        w, e = C.c_float(), C.c_float()
        L.dta_xchg_last_timing(ex._h, C.byref(w), C.byref(e))
        ws.append(w.value); es.append(e.value)
    ex.check()
    if rank == 0:
        es = sorted(es[5:]); ws = sorted(ws[5:])
        print(f"world {world} wgs {wgs} {mode}: exchange median {es[len(es)//2]:.1f} us (min {es[0]:.1f}), wait-for-ranks median {ws[len(ws)//2]:.1f} us", flush=True)
    dist.barrier(); ex.close(); dist.destroy_process_group()

def fp():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p

if __name__ == "__main__":
    for world, wgs in ((1, 256), (1, 128), (2, 128), (4, 64), (8, 32), (8, 110)):
        for mode in ("adam", "allreduce"):
            try:
                mp.spawn(worker, args=(world, fp(), wgs, mode), nprocs=world, join=True)
            except Exception as e:
                print("FAILED", world, wgs, str(e)[-300:])
