"""Peer gradient exchange (csrc/xchg.hip, dta_xchg_*): N processes sharing ONE GPU map each other's gradient buffers
through HIP IPC and run the fused all-reduce + Adam launch.  Expected values are computed on the host with the same
float32 operation order (sum in rank order, then the Adam formula of torch.optim.Adam, reference src/main.py:136)."""
import datetime
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N = 900_788          # the Hang2020(369, 200) flat buffer is about this long; not a multiple of the world sizes


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _grad(rank, step, n):
    i = np.arange(n, dtype=np.int64)
    return (((i * 2654435761 + rank * 40503 + step * 9973) % 2001 - 1000).astype(np.float32) / 1024.0)


def _alpha_grad(rank, step):
    return 1e-3 * (rank + 1) * step + 1e-11        # (not representable in float32: the slot carries its rounding)


def _adam_np(p, m, v, g, step, lr, b1, b2, eps, scale):
    f = np.float32
    g = (g * f(scale)).astype(f)
    m = (f(b1) * m + (f(1) - f(b1)) * g).astype(f)
    v = (f(b2) * v + (f(1) - f(b2)) * g * g).astype(f)
    bc1, bc2 = f(1.0 - b1 ** step), f(1.0 - b2 ** step)
    ss, rbc2 = f(lr) / bc1, f(1.0) / np.sqrt(bc2, dtype=f)
    p = (p - ss * (m / (np.sqrt(v, dtype=f) * rbc2 + f(eps)))).astype(f)
    return p, m, v


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    torch.cuda.set_device(0)
    return dist


def _worker(rank, world, port, out):
    dist = _init(rank, world, port)
    from deeptreeattention_amd.dist import PeerExchange
    dev = torch.device("cuda:0")
    ex = PeerExchange(N, timeout_s=20.0, max_workgroups=32)
    n = ex.capacity
    res = {}
    # plain all-reduce, twice (flag reuse), the second time with a straggler
    for step in range(2):
        ex.grad.copy_(torch.from_numpy(np.resize(_grad(rank, step, N), n)).to(dev))
        if step == 1 and rank == world - 1:
            torch.cuda.synchronize()
            time.sleep(0.3)
        ex.allreduce()
        torch.cuda.synchronize()
        ex.check()
        res[f"sum{step}"] = ex.grad.cpu().numpy()[:N].copy()
    # fused all-reduce + Adam with alpha's slot, three steps; rank-dependent gradients, identical start
    p = torch.from_numpy(np.resize(_grad(7, 7, N), n)).to(dev)
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    alpha = torch.full((), 0.5, dtype=torch.float64, device=dev)
    am = torch.zeros((), dtype=torch.float64, device=dev)
    av = torch.zeros((), dtype=torch.float64, device=dev)
    ag = torch.zeros((), dtype=torch.float64, device=dev)
    slot = 12345
    for step in range(1, 4):
        ex.grad.copy_(torch.from_numpy(np.resize(_grad(rank, 10 + step, N), n)).to(dev))
        ag.fill_(_alpha_grad(rank, step))          # this rank's float64 d(alpha): enters the sum through the slot
        ex.adam_step(p, m, v, alpha, ag, slot, am, av, step, 1e-3, (0.9, 0.999), 1e-8, zero_grad=(step != 3))
    torch.cuda.synchronize()
    ex.check()
    res["p"], res["m"], res["v"] = p.cpu().numpy()[:N], m.cpu().numpy()[:N], v.cpu().numpy()[:N]
    res["alpha"], res["ag"] = float(alpha), float(ag)
    res["g_last"] = ex.grad.cpu().numpy()[:N].copy()
    out[rank] = res
    ex.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_peer_allreduce_and_adam(world):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for step in range(2):
        want = _grad(0, step, N)
        for r in range(1, world):
            want = want + _grad(r, step, N)            # float32, rank order
        for r in range(world):
            assert np.array_equal(out[r][f"sum{step}"], want), (step, r)
    p = _grad(7, 7, N)
    m = np.zeros(N, np.float32)
    v = np.zeros(N, np.float32)
    alpha, am, av = 0.5, 0.0, 0.0
    for step in range(1, 4):
        g = _grad(0, 10 + step, N)
        for r in range(1, world):
            g = g + _grad(r, 10 + step, N)
        gsum = np.float32(_alpha_grad(0, step))
        for r in range(1, world):
            gsum = np.float32(gsum + np.float32(_alpha_grad(r, step)))
        g[12345] = gsum                                # the slot's gradient-buffer content is replaced by d(alpha)
        p, m, v = _adam_np(p, m, v, g, step, 1e-3, 0.9, 0.999, 1e-8, 1.0 / world)
        ga = float(gsum) / world
        am = 0.9 * am + 0.1 * ga
        av = 0.999 * av + 0.001 * ga * ga
        alpha -= (1e-3 / (1 - 0.9 ** step)) * (am / (np.sqrt(av) / np.sqrt(1 - 0.999 ** step) + 1e-8))
    for r in range(world):
        assert np.array_equal(out[r]["p"], out[0]["p"]) and np.array_equal(out[r]["m"], out[0]["m"])   # bit-identical replicas
        np.testing.assert_allclose(out[r]["p"], p, rtol=0, atol=3e-6)
        np.testing.assert_allclose(out[r]["m"], m, rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(out[r]["v"], v, rtol=1e-5, atol=1e-12)
        assert abs(out[r]["alpha"] - alpha) < 1e-9
        assert np.array_equal(out[r]["g_last"], g)                # zero_grad = 0 keeps the summed gradient
        assert abs(out[r]["ag"] - float(g[12345])) < 1e-12


def test_single_process_exchange_is_plain_adam():
    from deeptreeattention_amd.dist import PeerExchange
    dev = torch.device("cuda:0")
    ex = PeerExchange(1001)
    n = ex.capacity
    g = torch.randn(n, device=dev)
    ex.grad.copy_(g)
    p = torch.randn(n, device=dev)
    p0 = p.clone()
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    ex.adam_step(p, m, v, None, None, -1, None, None, 1, 1e-3, (0.9, 0.999), 1e-8, zero_grad=True)
    torch.cuda.synchronize()
    ex.check()
    ref = torch.nn.Parameter(p0.clone())
    ref.grad = g.clone()
    torch.optim.Adam([ref], lr=1e-3).step()
    assert torch.allclose(p, ref.data, rtol=0, atol=2e-6)
    assert float(ex.grad.abs().max()) == 0.0
    ex.close()


def _lonely_worker(rank, world, port, out):
    dist = _init(rank, world, port)
    from deeptreeattention_amd.dist import PeerExchange
    ex = PeerExchange(4096, timeout_s=0.25)
    if rank == 0:
        ex.allreduce()                   # rank 1 never joins this step
        torch.cuda.synchronize()
        try:
            ex.check()
            out[0] = "no error"
        except RuntimeError as e:
            out[0] = str(e)
        # the abort is STICKY: a later fused-Adam launch of the same exchange returns at once and applies nothing
        # (a replica must not train on past a failed exchange)
        n = ex.capacity
        ex.grad.fill_(1.0)
        p = torch.ones(n, device="cuda")
        m = torch.zeros(n, device="cuda")
        v = torch.zeros(n, device="cuda")
        t1 = time.time()
        ex.adam_step(p, m, v, None, None, -1, None, None, 1, 1e-3, (0.9, 0.999), 1e-8, zero_grad=True)
        torch.cuda.synchronize()
        out["sticky"] = (time.time() - t1, float(p.min()), float(m.abs().max()), float(ex.grad.min()))
    dist.barrier()
    ex.close()
    dist.destroy_process_group()


def test_missing_peer_times_out_instead_of_hanging():
    mgr = mp.Manager()
    out = mgr.dict()
    t0 = time.time()
    mp.spawn(_lonely_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert "timed out" in out[0] and "rank 1" in out[0], out[0]
    assert time.time() - t0 < 60
    dt, pmin, mmax, gmin = out["sticky"]
    assert dt < 0.2 and pmin == 1.0 and mmax == 0.0 and gmin == 1.0, out["sticky"]      # nothing waited, nothing applied


# ---- world = 8 on the REAL Hang2020(369, 200) flat layout (round-4 review: world 8 had never executed in any form, and no
#      shard arithmetic -- head / tail split, per-rank shard lengths -- had been exercised at 8 shards) ---------------------
def _real_layout():
    """(n, split) of the flat buffers a FusedTrainer builds for Hang2020(369, 200): head = everything but the first conv's
    weights (+ alpha's exchange slot), tail = the two first-conv weights (engine.FusedTrainer.bucket_sizes)."""
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.engine import FusedTrainer
    head, tail = FusedTrainer.bucket_sizes(H.Hang2020(369, 200))
    return head + tail, head


def _w8_worker(rank, world, port, out):
    dist = _init(rank, world, port)
    from deeptreeattention_amd.dist import PeerExchange
    dev = torch.device("cuda:0")
    n, split = _real_layout()
    ex = PeerExchange(n, timeout_s=60.0, max_workgroups=16, split=split)
    assert ex.capacity == n and n % 4 == 0 and split % 4 == 0
    res = {"n": n, "split": split}
    slot = split - 4                               # alpha's exchange slot: the head's last quad (FusedTrainer.alpha_slot_off)

    def fill(step):
        ex.grad.copy_(torch.from_numpy(_grad(rank, step, n)).to(dev))

    # (1) plain form of the two-segment buffer: one launch sums head and tail
    fill(0)
    ex.allreduce()
    torch.cuda.synchronize(); ex.check()
    res["sum_plain"] = ex.grad.cpu().numpy().copy()
    # (2) overlapped form: the head's reduce-scatter as its own launch (the device code of the weight-gradient launch's side
    #     workgroups), then the all-reduce launch that finds the head summed
    fill(1)
    ex.reduce_head()
    ex.allreduce()
    torch.cuda.synchronize(); ex.check()
    res["sum_overlap"] = ex.grad.cpu().numpy().copy()
    # (3) three fused Adam steps from identical parameters: overlapped, plain, overlapped; alpha through its slot; the
    #     gradients are cleared by steps 1 and 2 (the next step's buffer must start from zeros) and kept by step 3
    p = torch.from_numpy(_grad(7, 7, n)).to(dev)
    m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    alpha = torch.full((), 0.5, dtype=torch.float64, device=dev)
    am = torch.zeros((), dtype=torch.float64, device=dev); av = torch.zeros((), dtype=torch.float64, device=dev)
    ag = torch.zeros((), dtype=torch.float64, device=dev)
    for step in range(1, 4):
        if step > 1:
            res[f"cleared{step}"] = float(ex.grad.abs().max())      # what the previous step left behind
        fill(10 + step)
        ag.fill_(_alpha_grad(rank, step))
        if step != 2:
            ex.reduce_head(ag, slot)
        ex.adam_step(p, m, v, alpha, ag, slot, am, av, step, 1e-3, (0.9, 0.999), 1e-8, zero_grad=(step != 3))
        torch.cuda.synchronize(); ex.check()
    res["p"], res["m"], res["v"] = p.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy()
    res["alpha"] = float(alpha)
    res["g_last"] = ex.grad.cpu().numpy().copy()
    out[rank] = res
    ex.close()
    dist.destroy_process_group()


def test_world_8_on_the_real_hang2020_flat_layout():
    """Eight processes (sharing the one GPU: real IPC mappings, real flag protocol) exchange the 900,7xx-float buffer of
    Hang2020(369, 200) cut into head [0, split) and tail: the sum equals the host's float32 sum in rank order on every rank,
    plain and overlapped; three fused Adam steps give bit-identical replicas equal to the host's Adam; steps that clear
    leave exact zeros behind."""
    world = 8
    n, split = _real_layout()
    assert 900_000 < n < 901_000 and 0 < split < n
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_w8_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert sorted(out.keys()) == list(range(world))
    for key, step in (("sum_plain", 0), ("sum_overlap", 1)):
        want = _grad(0, step, n)
        for r in range(1, world):
            want = want + _grad(r, step, n)            # float32, rank order
        for r in range(world):
            assert np.array_equal(out[r][key], want), (key, r, int((out[r][key] != want).sum()))
    p = _grad(7, 7, n)
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    alpha, am, av = 0.5, 0.0, 0.0
    slot = split - 4
    for step in range(1, 4):
        g = _grad(0, 10 + step, n)
        for r in range(1, world):
            g = g + _grad(r, 10 + step, n)
        gsum = np.float32(_alpha_grad(0, step))
        for r in range(1, world):
            gsum = np.float32(gsum + np.float32(_alpha_grad(r, step)))
        g[slot] = gsum
        p, m, v = _adam_np(p, m, v, g, step, 1e-3, 0.9, 0.999, 1e-8, 1.0 / world)
        ga = float(gsum) / world
        am = 0.9 * am + 0.1 * ga
        av = 0.999 * av + 0.001 * ga * ga
        alpha -= (1e-3 / (1 - 0.9 ** step)) * (am / (np.sqrt(av) / np.sqrt(1 - 0.999 ** step) + 1e-8))
    for r in range(world):
        assert out[r]["n"] == n and out[r]["split"] == split
        for k in ("p", "m", "v"):
            assert np.array_equal(out[r][k], out[0][k]), (k, r)           # bit-identical replicas
        np.testing.assert_allclose(out[r]["p"], p, rtol=0, atol=3e-6)
        np.testing.assert_allclose(out[r]["m"], m, rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(out[r]["v"], v, rtol=1e-5, atol=1e-12)
        assert abs(out[r]["alpha"] - alpha) < 1e-9
        assert out[r]["cleared2"] == 0.0 and out[r]["cleared3"] == 0.0    # a clearing step leaves zeros for the next backward
        assert np.array_equal(out[r]["g_last"], g)                        # zero_grad = 0 keeps the summed gradient
