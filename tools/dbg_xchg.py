import os, sys, time, datetime, socket
import numpy as np, torch, ctypes as C
import torch.multiprocessing as mp
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))

def worker(rank, world, port, wgs):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]=str(port); os.environ.setdefault("GLOO_SOCKET_IFNAME","lo")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    torch.cuda.set_device(0)
    from deeptreeattention_amd.dist import PeerExchange
    from deeptreeattention_amd import _lib
    ex = PeerExchange(900788, timeout_s=4.0, max_workgroups=wgs)
    L=_lib.lib()
    for step in range(4):
        ex.grad.fill_(float(rank+1))
        torch.cuda.synchronize(); dist.barrier()
        if step==2 and rank==world-1: time.sleep(0.2)
        t0=time.time(); ex.allreduce(); torch.cuda.synchronize(); dt=time.time()-t0
        st=L.dta_xchg_status(ex._h)
        print(f"world {world} wgs {wgs} rank {rank} step {step} dt {dt*1e3:.2f} ms status {st:#x} val {float(ex.grad[5])}", flush=True)
        if st: break
    dist.barrier(); ex.close(); dist.destroy_process_group()

def fp():
    s=socket.socket(); s.bind(("127.0.0.1",0)); p=s.getsockname()[1]; s.close(); return p
if __name__=="__main__":
    for world, wgs in ((1,256),(2,256),(4,64),(4,256)):
        try: mp.spawn(worker, args=(world, fp(), wgs), nprocs=world, join=True)
        except Exception as e: print("FAILED", world, wgs, str(e)[-300:])
