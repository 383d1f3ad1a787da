"""The bench.py output contract: ONE JSON line with the driver's keys plus the `roofline` and `cpu_baseline` objects.
CPU: the committed artifact under profiles/ has the full schema.  GPU: a short live run prints a conforming line."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}
ROOF = {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def _check(line, want_cpu):
    d = json.loads(line)
    assert KEYS <= set(d), KEYS - set(d)
    assert d["unit"] == "patches/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert "model" not in d["config"]
    r = d["roofline"]
    assert ROOF <= set(r), ROOF - set(r)
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(d["value"] - d["steps"] * d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-3 * d["value"]
    if want_cpu:
        c = d["cpu_baseline"]
        assert {"value", "unit", "cores", "kind", "sample"} <= set(c) and c["kind"] in ("port", "reference")
    return d


@pytest.mark.parametrize("name", ["r01_bench_bf16_default.json", "r02_bench_bf16_default.json"])
def test_committed_bench_artifact_has_the_contract_schema(name):
    d = _check(open(os.path.join(REPO, "profiles", name)).read(), True)
    assert d["n_gpus"] == 1 and d["dtype"] == "bf16"
    if name.startswith("r02"):      # round 2: both first-conv rooflines, the step's, the steady state, the core count
        assert d["roofline_mfma"]["bound"] == "mfma" and d["roofline"]["bound"] == "hbm"
        assert d["step_roofline"]["algorithmic_flop_per_patch"] == 154486824
        assert d["steady_state"]["steps"] >= 200 and d["cpu_baseline"]["cores"] >= d["cpu_baseline"]["threads"]


@pytest.mark.gpu
def test_bench_prints_one_conforming_json_line():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "4", "--warmup", "2", "--site-stride", "2",
                          "--no-cpu-baseline", "--no-side"], capture_output=True, text=True, timeout=240, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = _check(lines[0], False)
    assert d["steps"] == 4 and d["warmup"] == 2 and d["n_gpus"] == 1
    # both first-conv kernels timed in the same run, the whole step priced, a >= 200-step median beside the K steps
    m = d["roofline_mfma"]
    assert ROOF <= set(m) and m["bound"] == "mfma" and m["launches"] == 4 and d["roofline"]["launches"] == 2      # (the reported kernel: every 2nd timed step)
    sr = d["step_roofline"]
    assert abs(sr["frac"] - sr["achieved"] / sr["peak"]) < 1e-3 and sr["algorithmic_flop_per_patch"] == 154486824
    ss = d["steady_state"]
    assert ss["steps"] >= 200 and 0.5 * d["ms_per_step"] < ss["median_ms_per_step"] < 2 * d["ms_per_step"]


@pytest.mark.gpu
def test_bench_two_ranks_print_one_json_line():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), with both ranks on
    the one GPU of the test box and gloo standing in for RCCL (DTA_BENCH_BACKEND=gloo, development only): exercises
    the rendezvous, the barriers, the two-bucket overlapped exchange, the max-over-ranks timing and rank 0's single
    JSON line.  Nothing about its speed is meaningful."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, DTA_BENCH_BACKEND="gloo", GLOO_SOCKET_IFNAME="lo", MASTER_ADDR="127.0.0.1")
    for flags, ncoll in (([], 2), (["--no-overlap"], 1)):
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                              "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py"),
                              "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "64", "--steady-steps", "0",
                              "--exchange", "torch"]
                             + flags, capture_output=True, text=True, timeout=420, cwd=REPO, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2"
        assert d["config"]["collectives_per_step"] == ncoll and "cpu_baseline" not in d
        assert d["config"]["ranks_seen"] == 2


@pytest.mark.gpu
def test_plain_bench_gpus_2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` started WITHOUT a launcher (no RANK in the environment) must spawn its ranks itself and
    still print exactly one JSON line: the first 8-GPU run of the driver must not fail on how it was started.  Both ranks
    share the test box's one GPU (DTA_BENCH_BACKEND=gloo, development only); the default exchange is the peer exchange
    (its crash-isolated probe passes between two processes of one device), overlapped with the backward."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(DTA_BENCH_BACKEND="gloo", GLOO_SOCKET_IFNAME="lo")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--batch", "64", "--steady-steps", "0"], capture_output=True, text=True, timeout=420, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["exchange"] == "peer" and d["config"]["exchange_fallbacks"] == []


@pytest.mark.gpu
def test_default_bench_line_carries_the_side_workloads():
    """The one-GPU driver line times, after the contract's region, the reference's own precision (fp32), BASELINE
    configs[4] (ensemble24) and the unchanged-reference-step plugin path (module_path), 20 steps each."""
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "4", "--warmup", "2", "--steady-steps", "20",
                          "--tile-steps", "0", "--no-cpu-baseline", "--prime-seconds", "0.2"], capture_output=True, text=True,
                         timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")][0])
    for k in ("fp32", "ensemble24", "metadata", "module_path"):
        assert k in d and "error" not in d[k], d.get(k)
    assert d["fp32"]["dtype"] == "fp32" and d["fp32"]["ms_per_step"] > d["ms_per_step"]
    assert d["metadata"]["sites"] == 23 and 0 < d["metadata"]["ms_per_step"] < 3 * d["ms_per_step"]
    assert d["ensemble24"]["config"]["crop"] == 24 and d["ensemble24"]["roofline"]["frac"] < d["ensemble24"]["roofline"]["frac_with_byproduct"]
    mp = d["module_path"]
    assert mp["hang2020_dta_adam_ms_per_step"] < mp["hang2020_torch_adam_ms_per_step"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["frac"] < r["frac_with_byproduct"]


@pytest.mark.gpu
def test_bench_two_ranks_fall_back_when_an_exchange_fails():
    """bench.py tries the gradient exchange for a few untimed steps and, if ANY rank fails, every rank drops it and takes
    the next of peer -> rccl -> torch.  Here rank 1 alone fails the first choice and both ranks fail the second (test
    hook DTA_BENCH_FAIL_EXCHANGE): the run must still end with one JSON line, on the torch exchange, saying what it
    gave up."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, DTA_BENCH_BACKEND="gloo", GLOO_SOCKET_IFNAME="lo", MASTER_ADDR="127.0.0.1",
               DTA_BENCH_FAIL_EXCHANGE="None@1,rccl")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "64", "--steady-steps", "0"],
                         capture_output=True, text=True, timeout=420, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["exchange"] == "torch"
    fb = d["config"]["exchange_fallbacks"]
    assert len(fb) == 2 and fb[1]["exchange"] == "rccl"


@pytest.mark.gpu
def test_plain_bench_gpus_8_spawns_eight_ranks():
    """`DTA_BENCH_BACKEND=gloo python bench.py --gpus 8 --steps 3` (round-4 review: world = 8 had never executed in any form):
    bench.py spawns its eight ranks itself, they share the test box's one GPU (development mode: never a measured
    configuration), rendezvous on 127.0.0.1, choose the exchange collectively (the peer exchange: its probe passes between
    processes of one device), run the overlapped step at the REAL 369-band / 200-class flat layout with 8 shards, and rank 0
    prints exactly one JSON line whose process group spanned 8 ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(DTA_BENCH_BACKEND="gloo", GLOO_SOCKET_IFNAME="lo")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                          "--batch", "32", "--steady-steps", "0"], capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["ranks_seen"] == 8 and d["config"]["parallelism"] == "dp8"
    assert d["config"]["global_batch"] == 256 and d["scaling"] == "weak"
    assert d["config"]["exchange"] == "peer" and d["config"]["exchange_fallbacks"] == [] and d["config"]["overlap_comm"] is True
    assert d["value"] > 0 and np.isfinite(d["final_loss"])
