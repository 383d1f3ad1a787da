"""Developer tool: fixed cost of each gradient exchange with ONE rank, for all three trainers (FusedTrainer: BASELINE
configs[1-2]; MetadataTrainer: configs[3]; EnsembleTrainer 3 x 369 x 24x24: configs[4]).  A one-rank RCCL process group with
DTA_FORCE_COLLECTIVES=1 still issues the step's exchange (nothing is on the wire), so the difference to the plain step is what
the exchange's launches / stream hops cost before a byte moves.  Same box, same process per trainer, 200-step medians.

    python tools/dp_one_rank.py [hang|metadata|ensemble24 ...]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STEPS = 200
OPTS = {"rccl": {"side_stream": True}}      # (rccl + overlap_comm=True: the side-stream two-bucket form)


def median_ms(step):
    for _ in range(30):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(STEPS)]
    for a, b in ev:
        a.record(); step(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return round(ms[len(ms) // 2], 4)


def build(kind, exchange, overlap):
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.engine import EnsembleTrainer, FusedTrainer, MetadataTrainer
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    g = torch.Generator(device=dev); g.manual_seed(2)
    if kind == "hang":
        m = H.Hang2020(369, 200, precision="bf16").to(dev).train()
        tr = FusedTrainer(m, lr=1e-4, loss_weight=torch.ones(200), exchange=exchange, overlap_comm=overlap, exchange_opts=OPTS.get(exchange))
        x = torch.rand(1024, 369, 11, 11, device=dev, generator=g); y = torch.randint(0, 200, (1024,), device=dev, generator=g)
        return tr, (lambda: tr.train_step(x, y)), bool(tr.overlap)
    if kind == "metadata":
        from deeptreeattention_amd.metadata import metadata_sensor_fusion
        m = metadata_sensor_fusion(bands=369, sites=23, classes=200, precision="bf16").to(dev).train()
        tr = MetadataTrainer(m, lr=1e-4, exchange=exchange, overlap_comm=overlap, exchange_opts=OPTS.get(exchange))
        x = torch.rand(1024, 369, 11, 11, device=dev, generator=g); y = torch.randint(0, 200, (1024,), device=dev, generator=g)
        site = torch.randint(0, 23, (1024,), device=dev, generator=g)
        return tr, (lambda: tr.train_step(x, site, y)), bool(tr.sensor.overlap)
    from deeptreeattention_amd.year import learned_ensemble
    m = learned_ensemble(3, 200, {"pretrain_state_dict": None, "bands": 369})
    for net in m.year_models:
        net.precision = "bf16"
    m = m.to(dev).train()
    tr = EnsembleTrainer(m, lr=1e-4, loss_weight=torch.ones(200), exchange=exchange, overlap_comm=overlap, exchange_opts=OPTS.get(exchange))
    xs = [torch.rand(256, 369, 24, 24, device=dev, generator=g) for _ in range(3)]
    y = torch.randint(0, 200, (256,), device=dev, generator=g)
    return tr, (lambda: tr.train_step(xs, y, present=[True, True, True])), bool(getattr(tr, "overlap_comm", False))


def main():
    kinds = sys.argv[1:] or ["hang", "metadata", "ensemble24"]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29581")
    torch.cuda.set_device(0)
    for kind in kinds:
        rows = {}
        os.environ.pop("DTA_FORCE_COLLECTIVES", None)
        tr, step, _ = build(kind, None, True)
        rows["plain (single process, no exchange)"] = median_ms(step)
        del tr, step
        torch.cuda.empty_cache()
        os.environ["DTA_FORCE_COLLECTIVES"] = "1"
        if not torch.distributed.is_initialized():
            torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        for name, exchange, overlap in (("peer, overlapped head segment", "peer", True), ("peer, one launch", "peer", False),
                                        ("rccl, side-stream two buckets", "rccl", True), ("rccl, one collective on the compute stream", "rccl", False),
                                        ("torch.distributed, two buckets", "torch", True), ("torch.distributed, one bucket", "torch", False)):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                tr, step, ov = build(kind, exchange, overlap)
                rows[f"{name} [overlap_comm={ov}]"] = median_ms(step)
            closer = tr.sensor if kind == "metadata" else tr
            closer.close()
            del tr, step
            torch.cuda.empty_cache()
        os.environ.pop("DTA_FORCE_COLLECTIVES", None)
        tr, step, _ = build(kind, None, True)
        rows["plain again"] = median_ms(step)
        del tr, step
        torch.cuda.empty_cache()
        base = rows["plain (single process, no exchange)"]
        print(json.dumps({"trainer": kind, "steps": STEPS, "median_ms_per_step": rows,
                          "overhead_us_vs_plain": {k: round((v - base) * 1e3, 1) for k, v in rows.items()}}), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
