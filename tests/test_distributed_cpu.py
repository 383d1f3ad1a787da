"""world_size-2 gloo tests (CPU) of the data-parallel host logic: gradient chunk all-reduce + averaging as the fused
trainer does it, parameter broadcast at start-up, and per-rank data sharding.  The HIP kernels need a GPU; what
is exercised here is deeptreeattention_amd.dist (the part of the N>1 path that is not a kernel)."""
import datetime
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # single node: no hostname resolution in the rendezvous
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    from deeptreeattention_amd.dist import GradSync, shard_seed, flat_layout
    torch.manual_seed(100 + rank)
    n, split = 1000, 700
    g = torch.randn(n, dtype=torch.float32)
    g[split - 1] = 0.0                     # the float64 alpha gradient's exchange slot inside the first bucket
    alpha_g = torch.tensor(float(rank + 1), dtype=torch.float64)
    mine = g.clone()
    sync = GradSync(world=world, group=None, side_stream=None)
    # two-phase reduction exactly as FusedTrainer.train_step issues it: two collectives, alpha rides in its slot
    sync.reduce_early(g[:split], alpha_g, g[split - 1:split])
    sync.reduce_late(g[split:])
    sync.finish()
    assert sync.collectives == 2
    gathered = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(gathered, mine)
    want = torch.stack(gathered).sum(0)
    want[split - 1] = sum(range(1, world + 1))
    ok_sum = torch.allclose(g, want, atol=1e-6)
    ok_alpha = abs(alpha_g.item() - sum(range(1, world + 1))) < 1e-12
    # single-bucket mode: ONE collective over the whole buffer
    g2 = mine.clone()
    a2 = torch.tensor(float(rank + 1), dtype=torch.float64)
    sync.reduce_all(g2, a2, g2[split - 1:split])
    sync.finish()
    assert sync.collectives == 3 and torch.allclose(g2, want, atol=1e-6) and abs(a2.item() - want[split - 1].item()) < 1e-12
    # averaging is applied by the optimizer kernel through grad_scale = 1/world
    ok_scale = abs(sync.grad_scale - 1.0 / world) < 1e-12
    # parameter broadcast
    p = torch.full((10,), float(rank))
    sync.broadcast([p], src=0)
    ok_bc = bool((p == 0).all())
    seeds = [shard_seed(1234, r) for r in range(world)]
    ok_seed = len(set(seeds)) == world
    out[rank] = (ok_sum, ok_alpha, ok_scale, ok_bc, ok_seed)
    dist.destroy_process_group()


def test_gradsync_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        assert all(out[r]), (r, out[r])


def test_flat_layout_puts_first_conv_last():
    from deeptreeattention_amd.dist import flat_layout
    sizes = [("a.conv1.conv_layer.weight", 100), ("a.conv1.conv_layer.bias", 4), ("a.fc", 30),
             ("b.conv1.conv_layer.weight", 50), ("b.x", 7)]
    order, split, total = flat_layout(sizes, late=lambda k: k.endswith("conv1.conv_layer.weight"))
    assert total == 191 and split == 41
    assert [k for k, _ in order[-2:]] == ["a.conv1.conv_layer.weight", "b.conv1.conv_layer.weight"]


def _ensemble_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # single node: no hostname resolution in the rendezvous
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    from deeptreeattention_amd.dist import GradSync
    # year 0 kept by both ranks, year 1 only by rank 1, year 2 by nobody.  As in EnsembleTrainer, each rank's 0/1 year
    # flags ride in slots of the first gradient bucket: the SUM all-reduce tells every rank which years were kept anywhere
    local = [True, rank == 1, False]
    sync = GradSync(world=world)
    # the year only rank 1 kept: rank 0 joins the same two-phase reduction with its zero-filled buffer
    g = torch.full((10,), 3.0) if local[1] else torch.zeros(10)
    head = torch.cat([g[:6], torch.tensor([1.0 if k else 0.0 for k in local])])
    tail = g[6:].clone()
    sync.reduce_early(head)
    sync.reduce_late(tail)
    sync.finish()
    anywhere = [f > 0 for f in head[6:].tolist()]
    out[rank] = (anywhere, head[:6].tolist() + tail.tolist(), sync.grad_scale)
    dist.destroy_process_group()


def test_ensemble_year_flags_and_zero_gradient_participation_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ensemble_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        anywhere, g, scale = out[r]
        assert anywhere == [True, True, False]
        assert g == [3.0] * 10 and scale == 0.5     # averaged gradient 1.5 on both ranks
