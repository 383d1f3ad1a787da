"""RCCL on the one GPU a test box has: a ONE-rank "nccl" process group with DTA_FORCE_COLLECTIVES=1 sends the step's two
gradient buckets (and the alpha slot) through real RCCL all-reduces on the side stream.  With one rank the sum is the
identity, so the updated weights must match the plain single-process step (alpha to fp32 rounding of its gradient:\nit travels in an fp32 slot); what this covers is the part
gloo cannot: RCCL initialisation with device_id, collectives enqueued from a non-default stream, the stream joins
around them, and bench.py's distributed branch on the nccl backend."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, torch
sys.path.insert(0, os.environ["DTA_ROOT"])
force = os.environ.get("DTA_FORCE_COLLECTIVES") == "1"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if force:
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from deeptreeattention_amd import Hang2020 as H
from deeptreeattention_amd.engine import FusedTrainer
torch.manual_seed(5)
m = H.Hang2020(40, 12, precision=os.environ["DTA_PREC"]).to(dev).train()
exch = os.environ.get("DTA_EXCH") or None
tr = FusedTrainer(m, lr=1e-3, overlap_comm=os.environ["DTA_OVERLAP"] == "1", exchange=exch,
                  exchange_opts={"side_stream": True} if exch == "rccl" else None)
g = torch.Generator(device=dev); g.manual_seed(7)
x = torch.rand(24, 40, 11, 11, device=dev, generator=g)
y = torch.randint(0, 12, (24,), device=dev, generator=g)
for _ in range(3):
    loss = tr.train_step(x, y)
torch.cuda.synchronize()
if force:
    assert tr.comm and tr.exchange == (exch or "torch")
    want = 0 if tr.exchange == "peer" else 3 * (2 if tr.overlap else 1)
    assert tr.sync.collectives == want, tr.sync.collectives
    tr.check_exchange()
    # broadcast + the collectives really went through RCCL
    assert torch.distributed.get_backend() == "nccl"
torch.save({k: v.cpu() for k, v in m.state_dict().items()}, os.environ["DTA_OUT"])
print("loss", float(loss))
if force:
    tr.close()
    torch.distributed.destroy_process_group()
"""


def _run(tmp_path, name, force, overlap, prec, exch=""):
    out = str(tmp_path / (name + ".pt"))
    env = dict(os.environ, DTA_ROOT=ROOT, DTA_OUT=out, DTA_OVERLAP="1" if overlap else "0", DTA_PREC=prec, DTA_EXCH=exch,
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DTA_FORCE_COLLECTIVES", None)
    if force:
        env["DTA_FORCE_COLLECTIVES"] = "1"
    r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return torch.load(out)


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_one_rank_rccl_step_matches_plain_step(tmp_path, overlap, prec):
    ref = _run(tmp_path, "plain", False, overlap, prec)
    got = _run(tmp_path, "rccl", True, overlap, prec)
    assert ref.keys() == got.keys()
    for k in ref:
        if k == "alpha":      # its float64 gradient rides through the exchange in an fp32 slot (dist.py)
            assert abs(float(ref[k]) - float(got[k])) < 1e-7
        else:                 # ... so later steps see an alpha that differs in the 8th digit
            assert torch.allclose(ref[k].float(), got[k].float(), rtol=1e-4, atol=1e-6), k


@pytest.mark.parametrize("exch", ["rccl", "peer"])
def test_one_rank_direct_exchanges_match_plain_step(tmp_path, exch):
    """ncclAllReduce called straight from librccl on the compute stream ("rccl") and the peer-exchange launch ("peer"),
    one rank each: the updated weights equal the plain single-process step."""
    ref = _run(tmp_path, "plain", False, False, "bf16")
    got = _run(tmp_path, exch, True, False, "bf16", exch)
    for k in ref:
        if k == "alpha":
            assert abs(float(ref[k]) - float(got[k])) < 1e-7
        else:
            assert torch.allclose(ref[k].float(), got[k].float(), rtol=1e-4, atol=1e-6), k


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_one_rank_rccl_side_stream_overlap_matches_plain_step(tmp_path, prec):
    """north_star's literal design: ncclAllReduce (librccl called directly) of the first bucket on a SIDE HIP stream, forked
    from the compute stream by an event after backward phase 1, running beside the first conv's weight gradient; the second
    bucket behind it; one event joins the compute stream before the optimizer (dist.RcclDirect.all_reduce_side / join: HIP
    streams and events only, no torch.distributed work objects).  One rank: the sums are identities, so the three steps
    must reproduce the plain single-process step; two collectives per step."""
    ref = _run(tmp_path, "plain", False, True, prec)
    got = _run(tmp_path, "rccl_overlap", True, True, prec, "rccl")
    for k in ref:
        if k == "alpha":
            assert abs(float(ref[k]) - float(got[k])) < 1e-7
        else:
            assert torch.allclose(ref[k].float(), got[k].float(), rtol=1e-4, atol=1e-6), k


@pytest.mark.parametrize("mode", ["plain", "rccl", "torch", "peer"])
def test_reruns_in_separate_processes_are_bit_identical(tmp_path, mode):
    """Three bf16 steps, twice, each in its own process: every tensor of the state dict has the same bits -- no float atomics
    on any path (alpha's fp32 exchange slot is ONE rounding of the finished float64 sum; the split-K sums meet in a fixed
    order), no dependence on allocation addresses."""
    force = mode != "plain"
    a = _run(tmp_path, mode + "_a", force, mode == "torch", "bf16", "" if mode in ("plain", "torch") else mode)
    b = _run(tmp_path, mode + "_b", force, mode == "torch", "bf16", "" if mode in ("plain", "torch") else mode)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_bench_runs_on_rccl_backend_with_one_rank():
    env = dict(os.environ, DTA_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29549", RANK="0",
               LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                        "--steady-steps", "20", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["roofline"]["frac"] > 0
