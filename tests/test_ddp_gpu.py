"""Two ranks on ONE GPU (gloo backend moves the CUDA gradient buffers through the host) exercising the real
data-parallel train step: backward phases 1/2, side-stream all-reduce, 1/world scaling in the Adam kernel.
Expected result: the oracle's Adam step on the MEAN of the two shards' gradients (per-rank BatchNorm statistics,
per-rank loss normalisation = DDP semantics)."""
import datetime
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

BANDS, CLASSES, B, SEED = 20, 7, 6, 13


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, overlap, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # single node: no hostname resolution in the rendezvous
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    from oracle import hang2020_np as O
    from oracle import prng
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.engine import FusedTrainer
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    p = O.init_params(O.hang2020_spec(BANDS, CLASSES), seed=SEED)
    m = H.Hang2020(BANDS, CLASSES)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in p.items()})
    if rank != 0:   # start-up broadcast must overwrite whatever a non-zero rank holds
        with torch.no_grad():
            for prm in m.parameters():
                prm.add_(0.5)
    m = m.to(dev).train()
    tr = FusedTrainer(m, lr=1e-3, overlap_comm=overlap, exchange="torch")
    assert tr.world == world
    x = prng.uniform01(100 + rank, 1, (B, BANDS, 11, 11))
    y = prng.randint(100 + rank, 2, (B,), CLASSES)
    before = tr.sync.collectives
    tr.train_step(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
    torch.cuda.synchronize()
    assert tr.sync.collectives - before == (2 if overlap else 1)      # alpha rides inside the fp32 buffer
    out[rank] = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,world", [(True, 2), (False, 2), (True, 4)])
def test_two_rank_train_step_matches_oracle(overlap, world):
    from conftest import rel_l2
    from oracle import hang2020_np as O
    from oracle import prng
    mgr = mp.Manager()
    out = mgr.dict()
    for attempt in range(2):      # a lost rendezvous (port taken between probe and bind) surfaces as a 90 s timeout: retry once
        try:
            mp.spawn(_worker, args=(world, _free_port(), overlap, out), nprocs=world, join=True)
            break
        except Exception:
            if attempt == 1:
                raise
    p = O.init_params(O.hang2020_spec(BANDS, CLASSES), seed=SEED)
    grads, upds = [], []
    for rank in range(world):
        x = prng.uniform01(100 + rank, 1, (B, BANDS, 11, 11))
        y = prng.randint(100 + rank, 2, (B,), CLASSES)
        logits, cache, upd = O.hang2020_fwd(p, x, True, np.float64)
        _, dl = O.weighted_cross_entropy(logits, y, np.ones(CLASSES, np.float32))
        grads.append(O.hang2020_bwd(p, cache, dl, np.float64))
        upds.append(upd)
    mean_g = {k: sum(np.asarray(g[k], np.float64) for g in grads) / world for k in grads[0]}
    want = O.adam_step(p, mean_g, {}, lr=1e-3)
    for k, v in want.items():
        if O.is_buffer(k) or k.endswith("conv_layer.bias"):
            continue
        for r in range(1, world):
            assert rel_l2(out[0][k], out[r][k]) < 1e-6, f"ranks diverged on {k}"  # replicas stay identical
        assert rel_l2(out[0][k], v) < 2e-3, k                                      # Adam on the averaged gradient
    for rank in range(world):   # BatchNorm buffers stay per-rank (no sync_batchnorm in the reference)
        for k, v in upds[rank].items():
            assert rel_l2(out[rank][k], v) < 2e-4, (rank, k)


def _ensemble_worker(rank, world, port, out):
    import datetime
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    from oracle import prng
    from deeptreeattention_amd.engine import EnsembleTrainer
    from deeptreeattention_amd.year import learned_ensemble
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    torch.manual_seed(5 + rank)                       # different start per rank: the start-up broadcast must fix it
    m = learned_ensemble(3, CLASSES, {"pretrain_state_dict": None, "bands": BANDS}).to(dev).train()
    tr = EnsembleTrainer(m, lr=1e-3, exchange="torch")
    losses = []
    for step in range(2):
        imgs = [torch.from_numpy(prng.uniform01(200 + 10 * step + rank, yy, (B, BANDS, 11, 11))).to(dev) for yy in range(3)]
        if rank == 0:
            imgs[1].zero_()                           # year 1 is missing on rank 0 only: it still takes part in the reduce
        if step == 1:
            imgs[2].zero_()                           # year 2 is missing everywhere in the second step: untouched
        y = torch.from_numpy(prng.randint(200 + rank, 2, (B,), CLASSES)).to(dev)
        losses.append(float(tr.train_step(imgs, y)))
    torch.cuda.synchronize()
    out[rank] = ({k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}, losses,
                 tr.step_counts(), tr.sync.collectives)
    dist.destroy_process_group()


def test_two_rank_ensemble_step_keeps_replicas_identical():
    """Year ensemble under data parallelism: a year missing on one rank only is still stepped everywhere (that rank
    contributes zero gradients), a year missing everywhere is left alone, and the replicas stay identical."""
    from conftest import rel_l2
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    for attempt in range(2):
        try:
            mp.spawn(_ensemble_worker, args=(world, _free_port(), out), nprocs=world, join=True)
            break
        except Exception:
            if attempt == 1:
                raise
    (sd0, l0, steps0, ncoll0), (sd1, l1, steps1, ncoll1) = out[0], out[1]
    assert steps0 == steps1 == [2, 2, 1]
    assert ncoll0 == ncoll1 and ncoll0 - 1 <= 2 * 2      # start-up broadcast aside: at most two collectives per step
    assert all(np.isfinite(l0)) and all(np.isfinite(l1))
    for k in sd0:
        if "running_" in k or "num_batches_tracked" in k:
            continue                                   # BatchNorm buffers are per rank (no sync_batchnorm)
        assert rel_l2(sd0[k], sd1[k]) < 1e-6, k
    assert int(sd0["year_models.1.conv1.bn1.num_batches_tracked"]) == 0      # rank 0 never ran year 1
    assert int(sd1["year_models.1.conv1.bn1.num_batches_tracked"]) == 2


def _metadata_worker(rank, world, port, out):
    import datetime
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    from oracle import prng
    from deeptreeattention_amd.engine import MetadataTrainer
    from deeptreeattention_amd.metadata import metadata_sensor_fusion
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    torch.manual_seed(9 + rank)
    m = metadata_sensor_fusion(bands=BANDS, sites=4, classes=CLASSES).to(dev).train()
    m.metadata_model.dropout.p = 0.0
    tr = MetadataTrainer(m, lr=1e-3, exchange="torch")
    for step in range(2):
        x = torch.from_numpy(prng.uniform01(300 + 10 * step + rank, 1, (B, BANDS, 11, 11))).to(dev)
        site = torch.from_numpy(prng.randint(300 + rank, 2, (B,), 4)).to(dev)
        y = torch.from_numpy(prng.randint(300 + rank, 3, (B,), CLASSES)).to(dev)
        loss = tr.train_step(x, site, y)
    torch.cuda.synchronize()
    out[rank] = ({k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}, float(loss))
    dist.destroy_process_group()


def test_two_rank_metadata_step_keeps_replicas_identical():
    from conftest import rel_l2
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    for attempt in range(2):
        try:
            mp.spawn(_metadata_worker, args=(world, _free_port(), out), nprocs=world, join=True)
            break
        except Exception:
            if attempt == 1:
                raise
    (sd0, l0), (sd1, l1) = out[0], out[1]
    assert np.isfinite(l0) and np.isfinite(l1)
    for k in sd0:
        if "running_" in k or "num_batches_tracked" in k:
            continue
        assert rel_l2(sd0[k], sd1[k]) < 1e-6, k


def _bb_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    from oracle import prng
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.engine import FusedTrainer
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    torch.manual_seed(3)
    m = H.Hang2020(BANDS, CLASSES).to(dev).train()
    tr = FusedTrainer(m, lr=1e-3, exchange="torch", broadcast_buffers=True)
    for step in range(3):
        x = torch.from_numpy(prng.uniform01(400 + 10 * step + rank, 1, (B, BANDS, 11, 11))).to(dev)
        y = torch.from_numpy(prng.randint(400 + rank, 2, (B,), CLASSES)).to(dev)
        tr.train_step(x, y)
    tr.sync_buffers()                      # what a checkpoint hook would call before saving on any rank
    torch.cuda.synchronize()
    out[rank] = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    dist.destroy_process_group()


def test_broadcast_buffers_option_gives_every_rank_rank0s_batchnorm_state():
    """Lightning DDP's broadcast_buffers semantics as an option: with it, the running statistics
    every rank holds are rank 0's; without it (default) they are per rank, which the other tests of this file pin."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bb_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for k in out[0]:
        assert np.array_equal(out[0][k], out[1][k]), k
    assert int(out[1]["spectral_network.conv1.bn1.num_batches_tracked"]) == 3
