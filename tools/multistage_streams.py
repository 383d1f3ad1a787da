"""Developer experiment: the reference's MultiStage step (5 levels x Y years, multi_stage.py:258-288) as five serial
EnsembleTrainer chains vs the same five chains on five HIP streams (one fork event, five join events).
    python tools/multistage_streams.py [B] [years] [steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeptreeattention_amd  # noqa: E402
from deeptreeattention_amd.engine import EnsembleTrainer  # noqa: E402
from deeptreeattention_amd.year import learned_ensemble  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
Y = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
deeptreeattention_amd.set_default_precision("bf16")
dev = torch.device("cuda:0")
cfg = {"pretrain_state_dict": None, "bands": 369}
classes = [2, 2, 12, 7, 5]
levels = [EnsembleTrainer(learned_ensemble(Y, c, cfg).to(dev).train(), lr=1e-4) for c in classes]
imgs = [[torch.rand(B, 369, 11, 11, device=dev) for _ in range(Y)] for _ in classes]
ys = [torch.randint(0, c, (B,), device=dev) for c in classes]
present = [True] * Y


def serial():
    return [t.train_step(x, y, present) for t, x, y in zip(levels, imgs, ys)]


streams = [torch.cuda.Stream(device=dev) for _ in classes]
fork = torch.cuda.Event()
joins = [torch.cuda.Event() for _ in classes]


def concurrent(nstreams=5):
    main = torch.cuda.current_stream()
    fork.record(main)
    out = []
    for i, (t, x, y) in enumerate(zip(levels, imgs, ys)):
        s = streams[i % nstreams]
        s.wait_event(fork)
        with torch.cuda.stream(s):
            out.append(t.train_step(x, y, present))
            joins[i].record(s)
    for i in range(len(levels)):
        main.wait_event(joins[i])
    return out


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {"B": B, "years": Y, "levels": len(classes), "steps": steps}
res["serial_ms"] = round(timed(serial, steps), 4)
for ns in (2, 3, 4, 5):
    res["streams_%d_ms" % ns] = round(timed(lambda: concurrent(ns), steps), 4)
res["serial_again_ms"] = round(timed(serial, steps), 4)
res["max_hw_queues_env"] = os.environ.get("GPU_MAX_HW_QUEUES")
print(json.dumps(res))

# ---- host side: how long does the enqueue of one serial step take when the GPU is not waited for? ----
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    serial()
host_ms = (time.perf_counter() - t0) / 20 * 1e3
torch.cuda.synchronize()
print(json.dumps({"host_enqueue_ms_per_serial_step(20 steps back to back, queue may fill)": round(host_ms, 4)}))

# ---- one host thread per level (ctypes releases the GIL inside the C-ABI calls), one stream per level ----
from concurrent.futures import ThreadPoolExecutor  # noqa: E402
pool = ThreadPoolExecutor(len(levels))


def _job(i):
    s = streams[i]
    with torch.cuda.stream(s):
        s.wait_event(fork)
        out = levels[i].train_step(imgs[i], ys[i], present)
        joins[i].record(s)
    return out


def threaded():
    main = torch.cuda.current_stream()
    fork.record(main)
    out = list(pool.map(_job, range(len(levels))))
    for j in joins:
        main.wait_event(j)
    return out


res2 = {"threaded_5_streams_ms": round(timed(threaded, steps), 4), "serial_ms": round(timed(serial, steps), 4)}
print(json.dumps(res2))
