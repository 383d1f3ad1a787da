#!/usr/bin/env python
"""bench.py - patches/sec of the full Hang2020 train step (forward + weighted CE + backward + Adam, + RCCL gradient
all-reduce when >1 GPU) at bands=369, 11x11, 200 classes (BASELINE.json metric / configs[1], configs[2]).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 50 --warmup 10

One process per GPU; per-GPU batch is fixed (weak scaling; 1024 per GPU = 8192 global on 8 GPUs).  Started WITHOUT a
launcher (`python bench.py --gpus N`, no RANK in the environment) it spawns its own N ranks under torch.distributed.run.
Synthetic patches (U[0,1), like the reference's min-max-scaled crops) are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.  `roofline.frac` prices SURVEY.md 8(d)'s per-patch bytes of the conv1 forward (the fp32 input:
182.9 MB per launch at B = 1024); with the half output: `frac_with_output`; with the bf16 tile by-product too: `frac_with_byproduct`.  The one-GPU line also carries, timed
after the contract's region (20 steps each): `fp32` (the reference's own precision), `ensemble24` (BASELINE configs[4]), `metadata` (BASELINE configs[3]),
`multistage` (the reference's train.py path: 5 levels x 3 years, one launch chain vs level by level) and `module_path` (the unchanged reference step on the plugin modules with optim.DtaAdam / optim.cross_entropy).  Inside the timed loop the two first-conv kernels are timed with HIP events recorded on their own
stream: `roofline` = the conv1 forward (the step's longest kernel; HBM-bound since it also converts the fp32 input and
emits the bf16 tiles), `roofline_mfma` = the conv1 weight gradient (the longest MFMA-bound kernel); `step_roofline`
prices the whole step's algorithmic FLOPs / bytes against the MI355X peaks.  `value` comes from the
K timed steps exactly as the driver's contract says; because K steps of 0.6 ms are a short region, `steady_state`
adds the median of >= 200 further steps timed one by one with HIP events.  `cpu_baseline` times the torch-eager port of
the reference step (oracle/hang2020_torch.py) on the host cores of the same box (rank 0, N=1 only, bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

BANDS, HW, CLASSES = 369, 11, 200
FLOP_PER_PATCH_STEP = 154_486_824       # torch FlopCounterMode on the reference, fwd+bwd
BYTES_PER_PATCH_STEP = 358_000          # compulsory HBM bytes per patch (fp32 input read twice)
CONV1_FLOP_PER_PATCH = 2 * 25_717_824   # both branches' first conv, forward (== its weight-gradient FLOPs)
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0                           # HBM3E, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024, help="patches per GPU per step")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--site", default="fwd0",
                    help="kernel site reported as `roofline` (the other first-conv kernel is reported as "
                         "`roofline_mfma` / `roofline_hbm` beside it): fwd0 = the first conv's forward (the longest "
                         "kernel of the step: it also converts the fp32 input and emits the bf16 tiles), wgrad0 = the "
                         "first conv's weight gradient (the longest MFMA-bound kernel)")
    ap.add_argument("--steady-steps", type=int, default=200,
                    help="extra steps timed one by one after the contract's K steps (median reported); 0 = skip")
    ap.add_argument("--other-steps", type=int, default=None,
                    help="steps after the timed region in which the OTHER first-conv kernel is event-timed (default min(steps, 50))")
    ap.add_argument("--tile-steps", type=int, default=100,
                    help="extra steps fed with pre-tiled bf16 input (reported under tile_input; 0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the side workloads (fp32 step, BASELINE configs[4] ensemble24, module-level plugin path) that the "
                         "default one-GPU line carries as sub-objects, 20 steps each after the contract's timed region")
    ap.add_argument("--cpu-batch", type=int, default=128)
    ap.add_argument("--cpu-seconds", type=float, default=24.0)
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--rccl-side-stream", action="store_true",
                    help="with --exchange rccl: two buckets on a side HIP stream overlapped with the backward (default: one collective on the compute stream)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "peer", "rccl", "torch"],
                    help="gradient exchange when data-parallel: peer = sum over ranks + Adam in one launch through "
                         "IPC-mapped peer memory (csrc/xchg.hip), rccl = one ncclAllReduce on the compute stream, torch = "
                         "torch.distributed buckets; auto = peer when its crash-isolated probe passes on every rank, else "
                         "rccl.  (A one-rank DTA_FORCE_COLLECTIVES run defaults to torch, the round-2 path.)")
    ap.add_argument("--workload", default="hang2020", choices=["hang2020", "ensemble24"],
                    help="hang2020 = BASELINE configs[1]/[2] (the headline); ensemble24 = BASELINE configs[4]: the year ensemble "
                         "(3 x spectral_network over 369-band 24x24 crops), a side workload with its own roofline object")
    ap.add_argument("--prime-seconds", type=float, default=0.75,
                    help="untimed train steps for this long BEFORE the W warmup steps: the first process on a freshly "
                         "started box runs its first ~0.2 s of GPU work at ramping clocks (measured: 0.73-0.78 ms per step "
                         "for the first 220 steps, 0.54 afterwards)")
    ap.add_argument("--site-stride", type=int, default=4,
                    help="inside the contract's timed steps the reported kernel is event-timed on every Nth step only (an "
                         "event pair costs ~5 us of stream time; measured: 11.6 us per step with a pair on every step)")
    ap.add_argument("--traffic-file", default=None, help="profiles/*_traffic_step.json whose PMC counters are quoted")
    return ap.parse_args()


def cpu_baseline(batch, seconds):
    """Reference step (stock torch ops + autograd + Adam) on the host cores; bounded sample.  oneDNN's scaling on
    this small model is poor beyond a few dozen threads, so a few (threads, batch) settings are tried for an equal
    slice of the time budget and the best is reported (with the threads actually used)."""
    from oracle import hang2020_np as O
    from oracle import hang2020_torch as TP
    from oracle import prng
    ncpu = os.cpu_count() or 1
    settings = []
    for thr in (16, 32, 64, ncpu // 2):
        thr = max(1, min(thr, ncpu))
        for b in (batch, 1024):
            if (thr, b) not in settings:
                settings.append((thr, b))
    base = O.init_params(O.hang2020_spec(BANDS, CLASSES), seed=1, randomize_bn=False)
    best = None
    tried = []
    budget = seconds / len(settings)
    for thr, b in settings:
        torch.set_num_threads(thr)
        p = TP.to_tensors(base)
        x = torch.from_numpy(prng.uniform01(0, 1, (b, BANDS, HW, HW)))
        y = torch.from_numpy(prng.randint(0, 2, (b,), CLASSES))
        step = TP.TrainStep(p, lr=1e-4, loss_weight=torch.ones(CLASSES))
        step(x, y)
        n, t0 = 0, time.perf_counter()
        while True:
            step(x, y)
            n += 1
            el = time.perf_counter() - t0
            if el >= budget or n >= 100:
                break
        rate = n * b / el
        tried.append(f"{thr}thr/b{b}:{rate:.0f}")
        if best is None or rate > best[0]:
            best = (rate, thr, b, n, el)
    rate, thr, b, n, el = best
    # `cores` (the contract's key) = the threads actually used; `hw_threads` = what os.cpu_count() reports for the box
    # (hardware threads, not physical cores)
    return {"value": round(rate, 1), "unit": "patches/s", "cores": thr, "threads": thr, "hw_threads": ncpu, "kind": "port",
            "sample": f"best of {len(settings)} settings ({', '.join(tried)} patches/s); reported: {n} train steps of "
                      f"batch {b}, fp32, torch {torch.__version__} eager on host CPU, {thr} of the box's {ncpu} "
                      f"hardware threads (oneDNN does not scale this small model further), {el:.1f} s"}


def main_ensemble24(a, emit=True):
    """BASELINE configs[4]: train step of the year ensemble (3 x spectral_network(369, 200) over 24x24 crops, mean of the
    last heads, weighted CE, backward, one Adam per year), bf16 convs / fp32 BN + loss, per-GPU batch 256 unless --batch
    says otherwise.  Single process (the data-parallel form of this step is covered by tests/test_ddp_gpu.py)."""
    from deeptreeattention_amd import _lib
    from deeptreeattention_amd.engine import EnsembleTrainer
    from deeptreeattention_amd.year import learned_ensemble
    import deeptreeattention_amd
    YEARS, CROP = 3, 24
    B = a.batch if a.batch != 1024 else 256
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    deeptreeattention_amd.set_default_precision(a.precision)
    torch.manual_seed(1234)
    m = learned_ensemble(YEARS, CLASSES, {"pretrain_state_dict": None, "bands": BANDS}).to(dev).train()
    tr = EnsembleTrainer(m, lr=1e-4, loss_weight=torch.ones(CLASSES))
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    imgs = [torch.rand(B, BANDS, CROP, CROP, device=dev, generator=g) for _ in range(YEARS)]
    y = torch.randint(0, CLASSES, (B,), device=dev, generator=g)
    present = [True] * YEARS                 # the loader knows which years exist (src/data.py zero-fills the others)
    L = _lib.lib()
    t_prime = time.perf_counter()
    while time.perf_counter() - t_prime < a.prime_seconds:       # clock ramp of a freshly started box (see --prime-seconds)
        for _ in range(20):
            tr.train_step(imgs, y, present)
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        tr.train_step(imgs, y, present)
    torch.cuda.synchronize()
    sites = {"fwd0": _lib.SITE_CONV_FWD, "wgrad0": _lib.SITE_CONV_WGRAD}
    L.dta_profile_set_stride(max(1, a.site_stride))
    L.dta_profile_enable(sites[a.site])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tr.train_step(imgs, y, present)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    buf = (C.c_float * 512)()
    n = L.dta_profile_collect_site(sites[a.site], buf, 512)
    ms = [buf[i] for i in range(max(n, 0))]
    L.dta_profile_enable(-1)
    L.dta_profile_set_stride(1)
    # algorithmic FLOPs (2 per MAC; conv layers only, as torch's FlopCounterMode counts the reference): per crop-year
    px = CROP * CROP
    conv_fwd = [2 * BANDS * 32 * 9 * px, 2 * 32 * 64 * 9 * px, 2 * 64 * 128 * 9 * (px // 4)]
    fwd = sum(conv_fwd) + 2 * 128 * CLASSES
    step_flop = fwd + sum(conv_fwd) + conv_fwd[1] + conv_fwd[2]        # + weight gradients + two input gradients
    bytes_step = 2 * BANDS * px * 4                                     # fp32 crop read by the forward and by conv1's weight gradient
    value = a.steps * B / el
    roof = None
    if ms:
        avg = sum(ms) / len(ms)
        flops = conv_fwd[0] * B * YEARS
        ach = flops / (avg * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": ("conv1 forward" if a.site == "fwd0" else "conv1 weight gradient") + " of the three years (one grouped launch)",
                "achieved": round(ach, 2), "peak": PEAK_TFLOPS[a.precision], "unit": "TFLOP/s",
                "frac": round(ach / PEAK_TFLOPS[a.precision], 4), "traffic": None, "avg_launch_ms": round(avg, 4),
                "launches": len(ms), "algorithmic_flop_per_launch": flops,
                "measured_in": f"HIP events around every {max(1, a.site_stride)}th launch inside the timed steps"}
        # HBM bytes per launch from committed PMC passes of THIS build (tools/run_profiles.sh: separate FETCH_SIZE / WRITE_SIZE
        # runs of this command; tools/step_traffic.py): null when the newest file was taken on another build of the kernels
        import glob
        cands = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_traffic_ensemble24.json")))
        roof["traffic_source"] = "none: no profiles/r*_traffic_ensemble24.json"
        if cands and B == 256 and a.precision == "bf16":
            tj = json.load(open(cands[-1]))
            build_id = L.dta_build_id().decode()
            if tj.get("library_build_id") == build_id:
                want = "k_conv3x3_bf16<3, 1, true" if a.site == "fwd0" else "k_conv_wgrad_bf16<4, 1"
                for row in tj.get("kernels", []):
                    if row["kernel"].startswith(want):
                        roof["traffic"] = row["hbm_bytes_per_launch"]
                roof["step_traffic"] = int(tj.get("hbm_mb_per_step", 0) * 1e6) or None
                roof["traffic_source"] = (f"PMC passes of this very build ({os.path.relpath(cands[-1], REPO)}, library_build_id {build_id}): "
                                          "2 x FETCH_SIZE + WRITE_SIZE in separate rocprofv3 runs")
            else:
                roof["traffic_source"] = (f"none: {os.path.relpath(cands[-1], REPO)} was measured on library build "
                                          f"{tj.get('library_build_id')}, this run is build {build_id}")
        if a.site == "fwd0" and a.precision == "bf16":
            # the bf16 first conv reads the fp32 crops itself and leaves the (haloed) bf16 tiles behind for the weight
            # gradient: per crop-year 369*576*4 B in, 384*676*2 B of tiles + 32*576*2 B of half output out -> HBM-bound
            # `frac` prices the COMPULSORY bytes only (the fp32 crop in, the half conv output out); the
            # bf16 tile by-product the kernel also writes (for its own weight gradient) is reported beside it
            withby = B * YEARS * (BANDS * px * 4 + 384 * (CROP + 2) * (CROP + 2) * 2 + 32 * px * 2)
            without = B * YEARS * (BANDS * px * 4 + 32 * px * 2)
            nbytes = B * YEARS * (BANDS * px * 4)          # the fp32 crops alone (as for the headline workload)
            gbs = nbytes / (avg * 1e-3) / 1e9
            roof.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(gbs / PEAK_HBM_GBS, 4), "algorithmic_bytes_per_launch": nbytes,
                         "frac_with_output": round(without / (avg * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                         "frac_with_byproduct": round(withby / (avg * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                         "bytes_per_launch_with_byproduct": withby, "mfma_tflops": round(ach, 2),
                         "kernel": "k_conv3x3_bf16<3,1,XN,6> (conv1 forward of the three years, one grouped launch; converts "
                                   "the fp32 crops and emits the bf16 tiles)"})
    out = {"metric": "crops/sec (train step) year-ensemble 3 x spectral_network 369-band 24x24", "value": round(value, 1),
           "unit": "crops/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(el / a.steps * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
           "config": {"workload": "year-ensemble train step (BASELINE configs[4]): 3 years x spectral_network(369, 200) on 24x24 crops, "
                                  "mean of last heads + weighted CE + bwd + Adam per year", "per_gpu_batch": B, "years": YEARS,
                      "crop": CROP, "parallelism": "dp1", "zero_year_test": "present flags from the loader (no host sync)"},
           "library_build_id": L.dta_build_id().decode(), "final_loss": round(float(loss), 5), "roofline": roof,
           "step_roofline": {"bound": "mfma", "achieved": round(value * YEARS * step_flop / 1e12, 2), "peak": PEAK_TFLOPS[a.precision],
                             "unit": "TFLOP/s", "frac": round(value * YEARS * step_flop / 1e12 / PEAK_TFLOPS[a.precision], 4),
                             "algorithmic_flop_per_crop_year": step_flop,
                             "hbm_gbs_algorithmic": round(value * YEARS * bytes_step / 1e9, 1),
                             "hbm_frac_algorithmic": round(value * YEARS * bytes_step / 1e9 / PEAK_HBM_GBS, 4)}}
    del tr, m, imgs
    torch.cuda.empty_cache()
    if emit:
        print(json.dumps(out), flush=True)
    return out


def side_fp32(a, dev, steps=20):
    """The reference's own precision: the same Hang2020 step with exact-fp32 MFMA contractions (precision='fp32')."""
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.engine import FusedTrainer
    torch.manual_seed(1234)
    m = H.Hang2020(BANDS, CLASSES, precision="fp32").to(dev).train()
    tr = FusedTrainer(m, lr=1e-4, loss_weight=torch.ones(CLASSES))
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    x = torch.rand(a.batch, BANDS, HW, HW, device=dev, generator=g)
    y = torch.randint(0, CLASSES, (a.batch,), device=dev, generator=g)
    for _ in range(5):
        tr.train_step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.train_step(x, y)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    out = {"dtype": "fp32", "steps": steps, "per_gpu_batch": a.batch, "ms_per_step": round(ms, 4),
           "patches_per_s": round(a.batch / ms * 1e3, 1),
           "mfma_frac_fp32_peak": round(a.batch / ms * 1e3 * FLOP_PER_PATCH_STEP / 1e12 / PEAK_TFLOPS["fp32"], 4),
           "final_loss": round(float(loss), 5),
           "note": "FusedTrainer, Hang2020(369, 200) precision='fp32' (v_mfma_f32_32x32x2_f32: exact fp32 products), wall clock "
                   "around the steps after 5 warm-up steps"}
    del tr, m, x
    torch.cuda.empty_cache()
    return out


def side_metadata(a, dev, steps=20, sites=23):
    """BASELINE configs[3]: the site-metadata fusion model's train step (reference src/models/metadata.py:26-63:
    ReLU(Linear(cat[site MLP, Hang2020 scores])), unweighted cross-entropy, Adam) through engine.MetadataTrainer: the HSI
    branch fused, the 16-wide site MLP + fusion layer a small torch graph joined at the (B, classes) scores."""
    from deeptreeattention_amd.engine import MetadataTrainer
    from deeptreeattention_amd.metadata import metadata_sensor_fusion
    torch.manual_seed(1234)
    m = metadata_sensor_fusion(bands=BANDS, sites=sites, classes=CLASSES, precision=a.precision).to(dev).train()
    tr = MetadataTrainer(m, lr=1e-4)
    g = torch.Generator(device=dev)
    g.manual_seed(98)
    x = torch.rand(a.batch, BANDS, HW, HW, device=dev, generator=g)
    site = torch.randint(0, sites, (a.batch,), device=dev, generator=g)
    y = torch.randint(0, CLASSES, (a.batch,), device=dev, generator=g)
    for _ in range(5):
        tr.train_step(x, site, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.train_step(x, site, y)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    out = {"dtype": a.precision, "steps": steps, "per_gpu_batch": a.batch, "sites": sites, "ms_per_step": round(ms, 4),
           "patches_per_s": round(a.batch / ms * 1e3, 1), "final_loss": round(float(loss), 5),
           "note": "MetadataTrainer(metadata_sensor_fusion(369, 23 sites, 200)): HSI branch = the fused Hang2020 step, site MLP + "
                   "fusion layer + loss = torch ops on (B, 200) tensors; wall clock around the steps after 5 warm-up steps"}
    del tr, m, x
    torch.cuda.empty_cache()
    return out


def side_multistage(a, dev, steps=20, years=3, classes=(2, 2, 12, 7, 5), lrs=(1e-6, 1e-6, 5e-6, 1e-4, 5e-6)):
    """The path the reference's train.py trains (train.py:75-100, src/models/multi_stage.py:41-66, :258-288): five levels x
    `years` spectral_network(369, classes_l) year ensembles on 11x11 crops, every level stepped on every batch (Lightning
    calls training_step once per optimizer), per-level class weights / Adam / learning rate (config.yml:62-66).  Timed at
    the reference's batch size (config.yml:58: 128) and at 1024: all levels x years networks as ONE launch chain
    (MultiStageTrainer.training_step_all, dta_multistage_*) against the level-by-level step."""
    from deeptreeattention_amd.engine import MultiStageTrainer
    from deeptreeattention_amd.year import learned_ensemble
    import deeptreeattention_amd
    deeptreeattention_amd.set_default_precision(a.precision)
    cfg = {"pretrain_state_dict": None, "bands": BANDS}
    torch.manual_seed(4321)
    tr = MultiStageTrainer([learned_ensemble(years, c, cfg).to(dev).train() for c in classes], list(lrs))
    out = {"dtype": a.precision, "steps": steps, "levels": len(classes), "years": years, "classes": list(classes),
           "note": "wall clock around the steps after 5 warm-up steps; present= flags (no year missing); batched = one launch chain "
                   "over the 15 networks, serial = one chain per level (what Lightning's per-optimizer loop does)"}
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    for B in (128, 1024):
        batch = [(None, {"HSI": [torch.rand(B, BANDS, HW, HW, device=dev, generator=g) for _ in range(years)]},
                  torch.randint(0, c, (B,), device=dev, generator=g)) for c in classes]
        present = [[True] * years] * len(classes)

        def timed(fn):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                losses = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3, losses
        ms_serial, _ = timed(lambda: [tr.training_step(batch, 0, l, present[l]) for l in range(len(classes))])
        ms_batched, losses = timed(lambda: tr.training_step_all(batch, 0, present))
        out["B%d" % B] = {"per_level_batch": B, "batched_ms_per_step": round(ms_batched, 4), "serial_ms_per_step": round(ms_serial, 4),
                          "speedup": round(ms_serial / ms_batched, 2), "one_chain": bool(tr.batched_last),
                          "crops_per_s": round(len(classes) * B / ms_batched * 1e3, 1),
                          "final_losses": [round(float(v), 5) for v in losses]}
        del batch
    # MultiStage.predict_step (multi_stage.py:306-318; config.yml:80 predict_batch_size 64): every level's ensemble on the SAME crops
    from deeptreeattention_amd.engine import MultiStagePredictor, Predictor
    for t in tr.levels:
        t.model.eval()
    per = [Predictor(t.model) for t in tr.levels]
    one = MultiStagePredictor([t.model for t in tr.levels])
    x = [torch.rand(64, BANDS, HW, HW, device=dev, generator=g) for _ in range(years)]

    def timed_p(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5 * steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (5 * steps) * 1e3
    ms_per, ms_one = timed_p(lambda: [p(x) for p in per]), timed_p(lambda: one(x))
    frz = MultiStagePredictor([t.model for t in tr.levels], frozen=True)      # trained weights: re-layouts kept in the workspace (DTA_REUSE_PACKED)
    ms_frz = timed_p(lambda: frz(x))
    out["predict_B64"] = {"batch": 64, "one_chain_ms": round(ms_one, 4), "per_level_ms": round(ms_per, 4), "speedup": round(ms_per / ms_one, 2),
                          "crops_per_s": round(64 / ms_one * 1e3, 1),
                          "one_chain_frozen_weights_ms": round(ms_frz, 4), "crops_per_s_frozen_weights": round(64 / ms_frz * 1e3, 1),
                          "note": "eval forward of all levels x years networks on the same crops + per-level softmax / top-2: one launch chain "
                                  "(MultiStagePredictor) vs a Predictor per level"}
    del tr, per, one, frz
    torch.cuda.empty_cache()
    return out


def side_module_path(a, dev, fused_ms, steps=20):
    """The UNCHANGED reference step (TreeModel.training_step: forward, F.cross_entropy, loss.backward(), optimizer.step(),
    src/main.py:71-80,135-149) on the plugin modules, with the two one-line swaps INTEGRATION.md section 1 names:
    optim.DtaAdam for optim.Adam and optim.cross_entropy for F.cross_entropy -- and with stock torch for both."""
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.optim import DtaAdam, cross_entropy
    from deeptreeattention_amd.year import learned_ensemble
    import deeptreeattention_amd
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    x = torch.rand(a.batch, BANDS, HW, HW, device=dev, generator=g)
    y = torch.randint(0, CLASSES, (a.batch,), device=dev, generator=g)
    w = torch.ones(CLASSES, device=dev)

    def timed(step, n):
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    out = {"steps": steps, "per_gpu_batch": a.batch, "dtype": a.precision}
    torch.manual_seed(1234)
    m = H.Hang2020(BANDS, CLASSES, precision=a.precision).to(dev).train()
    opt = DtaAdam(m.parameters(), lr=1e-4)

    def step_dta():
        opt.zero_grad()
        loss = cross_entropy(m(x), y, weight=w)
        loss.backward()
        opt.step()
    ms = timed(step_dta, steps)
    out["hang2020_dta_adam_ms_per_step"] = round(ms, 4)
    out["hang2020_vs_fused_trainer"] = round(ms / fused_ms, 3) if fused_ms else None
    opt.close()
    del opt, m
    torch.manual_seed(1234)
    m = H.Hang2020(BANDS, CLASSES, precision=a.precision).to(dev).train()
    topt = torch.optim.Adam(m.parameters(), lr=1e-4)

    def step_torch():
        topt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(m(x), y, weight=w)
        loss.backward()
        topt.step()
    out["hang2020_torch_adam_ms_per_step"] = round(timed(step_torch, steps), 4)
    del topt, m
    # the year ensemble the reference actually trains (multi_stage.py:277-288), 3 years of 11x11 patches, device-decided
    # missing years: module forward + optim.cross_entropy + DtaAdam's gated per-year steps
    deeptreeattention_amd.set_default_precision(a.precision)
    ens = learned_ensemble(3, CLASSES, {"pretrain_state_dict": None, "bands": BANDS}).to(dev).train()
    eopt = DtaAdam(ens.parameters(), lr=1e-4)
    imgs = [x, x.flip(0), x.flip(1)]

    def step_ens():
        eopt.zero_grad()
        loss = cross_entropy(ens(imgs), y, weight=w)
        loss.backward()
        eopt.step()
    out["ensemble3_11x11_dta_adam_ms_per_step"] = round(timed(step_ens, steps), 4)
    eopt.close()
    del eopt, ens, imgs, x
    deeptreeattention_amd.set_default_precision("fp32")
    torch.cuda.empty_cache()
    out["note"] = ("module-level plugin path (autograd.Function per network, gradients written in place into DtaAdam's flat "
                   "buffer); wall clock around the steps after 5 warm-up steps")
    return out


def main():
    a = parse()
    if a.workload == "ensemble24":
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
        return main_ensemble24(a)
    if a.gpus > 1 and "RANK" not in os.environ:
        # started as plain `python bench.py --gpus N`: become the launcher the contract names -- one process per GPU
        # under torch.distributed.run on 127.0.0.1 -- and hand its exit code on (rank 0 prints the one JSON line)
        import socket
        import subprocess
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch one process per GPU (plain `python bench.py "
                         f"--gpus N` does that itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    if os.environ.get("DTA_BENCH_BACKEND", "nccl") != "nccl":
        local = local % torch.cuda.device_count()   # development only: ranks may share a GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # DTA_FORCE_COLLECTIVES=1: a single rank still builds the RCCL process group and issues the step's collectives
    # (development: the only RCCL exercise a one-GPU box allows; never a measured configuration)
    dist_on = world > 1 or os.environ.get("DTA_FORCE_COLLECTIVES") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # "nccl" is RCCL on ROCm.  DTA_BENCH_BACKEND=gloo exists only to exercise this multi-rank code path with
        # several ranks sharing ONE GPU (development boxes); it is never a measured configuration.
        backend = os.environ.get("DTA_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)

    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd import _lib
    from deeptreeattention_amd.engine import FusedTrainer

    torch.manual_seed(1234)                      # same initial weights on every rank (then broadcast anyway)
    model = H.Hang2020(BANDS, CLASSES, precision=a.precision).to(dev)
    model.train()
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)                   # each rank owns a different shard of the global batch
    # distinct resident batches the steps cycle through: 4 x 183 MB of fp32 patches (> the 256 MB memory-side cache, so no
    # step finds its input there) -- and enough different patches (4096 per rank) that `final_loss` after the priming
    # steps is not the memorisation of two batches
    nb = 4
    xs = [torch.rand(a.batch, BANDS, HW, HW, device=dev, generator=g) for _ in range(nb)]
    ys = [torch.randint(0, CLASSES, (a.batch,), device=dev, generator=g) for _ in range(nb)]

    shared_gpu = dist_on and os.environ.get("DTA_BENCH_BACKEND", "nccl") != "nccl"

    def build_trainer(exchange):
        opts = None
        if exchange in (None, "peer"):
            opts = {"timeout_s": 30.0}
            if shared_gpu:          # development: ranks share one GPU and wait for each other in-kernel -> stay co-resident
                opts["max_workgroups"] = 32
        return FusedTrainer(model, lr=1e-4, loss_weight=torch.ones(CLASSES), overlap_comm=not a.no_overlap,
                            exchange=exchange, exchange_opts=opts)

    # Data-parallel: the exchange is tried for a few steps before anything is timed.  "auto" walks peer -> rccl -> torch:
    # the trainer's own choice first (its crash-isolated probe decides whether the peer exchange is usable at all), and
    # if ANY rank fails to build it or sees a step fail, every rank drops it and takes the next one -- a multi-GPU node
    # this code never ran on must still produce a measured line.  An explicit --exchange is taken as given.
    chain = [None, "rccl", "torch"] if a.exchange == "auto" else [a.exchange]
    if not dist_on:
        chain = chain[:1]
    trainer, fallbacks = None, []
    def agreed_failure(err):
        bad = torch.tensor([1.0 if err else 0.0], device=dev)
        torch.distributed.all_reduce(bad, op=torch.distributed.ReduceOp.MAX)
        return bad.item() != 0.0

    for k, choice in enumerate(chain):
        err = ""
        try:
            trainer = build_trainer(choice)
            hook = os.environ.get("DTA_BENCH_FAIL_EXCHANGE", "").split(",")      # (test hook: tests/test_bench_contract.py)
            if str(choice) in hook or f"{choice}@{rank}" in hook:
                raise RuntimeError("DTA_BENCH_FAIL_EXCHANGE")
        except Exception as e:      # noqa: BLE001 -- whatever went wrong, the ranks must agree on what happens next
            err = f"{type(e).__name__}: {e}"
        if not dist_on:
            if err:
                raise RuntimeError(err)
            break
        failed = agreed_failure(err)         # (a rank whose constructor failed cannot take part in the trial steps)
        if not failed:
            try:
                for i in range(4):
                    trainer.train_step(xs[i % nb], ys[i % nb])
                torch.cuda.synchronize()
                trainer.check_exchange()
            except Exception as e:      # noqa: BLE001
                err = f"{type(e).__name__}: {e}"
            failed = agreed_failure(err)
        if not failed:
            break
        fallbacks.append({"exchange": (trainer.exchange if trainer is not None else str(choice)), "error": err[:200]})
        if k + 1 == len(chain):
            raise RuntimeError(f"no gradient exchange works on this node: {fallbacks}")
        # teardown without the trainer's own collective close(): a rank whose constructor failed has nothing to close
        torch.cuda.synchronize()
        ex = getattr(trainer, "ex", None) if trainer is not None else None
        if ex is not None and ex._h is not None:
            trainer.flat_g = trainer.g_head = trainer.g_tail = None
            trainer._gview = {}
            ex.grad = None
        torch.distributed.barrier()                  # nobody unmaps while a peer may still read
        if ex is not None and ex._h is not None:
            ex._owner.destroy()
            ex._h = None
            trainer.ex = None
        trainer = None
        torch.manual_seed(1234)
        model = H.Hang2020(BANDS, CLASSES, precision=a.precision).to(dev)      # the trial steps moved the weights
        model.train()

    L = _lib.lib()
    SITE = {"fwd0": _lib.SITE_CONV_FWD, "wgrad0": _lib.SITE_CONV_WGRAD}
    ranks_seen = 1
    if dist_on:      # proof in the record that the process group really spans `world` ranks: an all-reduce of ones
        one = torch.ones(1, device=dev)
        torch.distributed.all_reduce(one)
        ranks_seen = int(one.item())

    def barrier():
        if dist_on:
            torch.distributed.barrier()

    primed = 0
    t_prime = time.perf_counter()
    while a.prime_seconds > 0:
        for i in range(50):
            trainer.train_step(xs[i % nb], ys[i % nb])
        torch.cuda.synchronize()
        primed += 50
        stop = time.perf_counter() - t_prime >= a.prime_seconds
        if dist_on:      # every rank must issue the same number of exchanges: any rank's "enough" ends it for all
            flag = torch.tensor([1.0 if stop else 0.0], device=dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            stop = flag.item() > 0.0
        if stop:
            break
    for i in range(a.warmup):
        trainer.train_step(xs[i % nb], ys[i % nb])
    torch.cuda.synchronize()
    barrier()
    # the kernel reported as `roofline` is timed inside the contract's region (one HIP-event pair per step, measured
    # cost 4-5 us per step and pair); the other first-conv kernel over further steps of the same run, so that the timed
    # region carries one pair, not two
    other = "wgrad0" if a.site == "fwd0" else "fwd0"
    if rank == 0:
        L.dta_profile_set_stride(max(1, a.site_stride))
        L.dta_profile_enable(SITE[a.site])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = trainer.train_step(xs[i % nb], ys[i % nb])
    torch.cuda.synchronize()
    el_own = time.perf_counter() - t0        # this rank's own steps, before it waits for the others: rank skew shows here
    barrier()
    el = time.perf_counter() - t0
    trainer.check_exchange()                 # a peer-exchange step that timed out waiting for a rank raises here
    rank_ms = {"min": round(el_own / a.steps * 1e3, 4), "max": round(el_own / a.steps * 1e3, 4)}
    if dist_on:
        t = torch.tensor([el, el_own, -el_own], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t[0].item())
        rank_ms = {"min": round(-float(t[2].item()) / a.steps * 1e3, 4), "max": round(float(t[1].item()) / a.steps * 1e3, 4)}
    final_loss = float(loss.item())

    def collect(site):
        buf = (C.c_float * 512)()
        n = L.dta_profile_collect_site(site, buf, 512)
        return [buf[i] for i in range(max(n, 0))]
    site_ms = {a.site: collect(SITE[a.site])} if rank == 0 else {}
    if rank == 0:
        L.dta_profile_enable(-1)
        L.dta_profile_set_stride(1)
    n_other = min(a.steps, 50) if a.other_steps is None else a.other_steps
    if rank == 0:
        L.dta_profile_enable(SITE[other])
    for i in range(n_other):                 # (every rank: the steps issue the collectives)
        trainer.train_step(xs[i % nb], ys[i % nb])
    torch.cuda.synchronize()
    barrier()
    if rank == 0:
        site_ms[other] = collect(SITE[other])
        L.dta_profile_enable(-1)

    # steady state beyond the contract's K steps: each step bracketed by its own HIP events (no host sync in between)
    steady = None
    if a.steady_steps > 0:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steady_steps)]
        for i, (e0, e1) in enumerate(evs):
            e0.record()
            trainer.train_step(xs[i % nb], ys[i % nb])
            e1.record()
        torch.cuda.synchronize()
        barrier()
        ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        steady = {"steps": a.steady_steps, "median_ms_per_step": round(ms[len(ms) // 2], 4), "min_ms_per_step": round(ms[0], 4),
                  "p90_ms_per_step": round(ms[int(0.9 * (len(ms) - 1))], 4),
                  "median_patches_per_s_per_gpu": round(a.batch / (ms[len(ms) // 2] * 1e-3), 1),
                  "note": "per-step HIP events on this rank's stream after the contract's timed region"}

    # the same step fed the way this framework's own loader delivers patches (preprocess -> the first conv's bf16 tiles,
    # SURVEY 8 row n2): the first conv then reads 2 bytes per value and writes no tile by-product.  A side number: the
    # headline above keeps the reference's input format (fp32 NCHW).
    tile_in = None
    if a.tile_steps > 0 and a.precision == "bf16" and not dist_on:      # (single process only: the steps would issue collectives)
        from deeptreeattention_amd.preprocess import PatchTiles
        nchunk = (BANDS + 15) // 16

        def as_tiles(x):
            pad = torch.zeros(a.batch, nchunk * 16, HW * HW, device=dev)
            pad[:, :BANDS] = x.reshape(a.batch, BANDS, HW * HW)
            t = pad.view(a.batch, nchunk, 16, HW * HW).permute(0, 1, 3, 2).contiguous().to(torch.bfloat16)
            return PatchTiles(t.view(torch.int16).reshape(-1), a.batch, BANDS, HW, HW)
        ts = [as_tiles(x) for x in xs]
        for i in range(5):
            trainer.train_step(ts[i % nb], ys[i % nb])
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.tile_steps)]
        for i, (e0, e1) in enumerate(evs):
            e0.record()
            trainer.train_step(ts[i % nb], ys[i % nb])
            e1.record()
        torch.cuda.synchronize()
        ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        tile_in = {"steps": a.tile_steps, "median_ms_per_step": round(ms[len(ms) // 2], 4),
                   "median_patches_per_s_per_gpu": round(a.batch / (ms[len(ms) // 2] * 1e-3), 1),
                   "note": "input handed over as the first conv's bf16 tiles (preprocess.preprocess_batch(tiles=True)); "
                           "per-step HIP events, rank 0, after the contract's timed region"}
        del ts
    barrier()

    if rank == 0:
        # HBM bytes per launch from the committed PMC passes (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 runs of this
        # very command; tools/step_traffic.py): a constant of the configuration it was measured on, not a live counter
        traffic = {}
        build_id = L.dta_build_id().decode()
        tpath = a.traffic_file
        if tpath is None:
            import glob
            cands = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_traffic_step.json")))
            tpath = cands[-1] if cands else ""
        tsrc = None
        if os.path.exists(tpath) and a.batch == 1024 and a.precision == "bf16":
            tj = json.load(open(tpath))
            if tj.get("library_build_id") == build_id:
                for row in tj.get("kernels", []):
                    if row["kernel"].startswith("k_conv3x3_bf16<2, 2, true"):
                        traffic["fwd0"] = row["hbm_bytes_per_launch"]
                    if row["kernel"].startswith("k_conv_wgrad_bf16<2, 2"):
                        traffic["wgrad0"] = row["hbm_bytes_per_launch"]
                traffic["step_total"] = int(tj.get("hbm_mb_per_step", 0) * 1e6) or None
                tsrc = (f"PMC passes of this very build ({os.path.relpath(tpath, REPO)}, library_build_id {build_id}), "
                        "batch 1024 bf16: 2 x FETCH_SIZE + WRITE_SIZE in separate rocprofv3 runs")
            else:
                # counters taken on another build of the kernels say nothing about this one: traffic stays null
                tsrc = (f"none: {os.path.relpath(tpath, REPO)} was measured on library build "
                        f"{tj.get('library_build_id')}, this run is build {build_id}")
        flops = CONV1_FLOP_PER_PATCH * a.batch
        roofs = {}
        if site_ms.get("wgrad0"):
            avg_ms = sum(site_ms["wgrad0"]) / len(site_ms["wgrad0"])
            ach = flops / (avg_ms * 1e-3) / 1e12
            roofs["wgrad0"] = {"bound": "mfma", "kernel": "k_conv_wgrad (conv1 weight gradient, both branches)",
                               "achieved": round(ach, 2), "peak": PEAK_TFLOPS[a.precision], "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_TFLOPS[a.precision], 4), "traffic": traffic.get("wgrad0"),
                               "traffic_source": tsrc, "avg_launch_ms": round(avg_ms, 4), "launches": len(site_ms["wgrad0"]),
                               "algorithmic_flop_per_launch": flops}
        if site_ms.get("fwd0"):
            avg_ms = sum(site_ms["fwd0"]) / len(site_ms["fwd0"])
            ach = flops / (avg_ms * 1e-3) / 1e12
            r = {"bound": "mfma", "kernel": "k_conv3x3 (conv1 forward, both branches)", "achieved": round(ach, 2),
                 "peak": PEAK_TFLOPS[a.precision], "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS[a.precision], 4),
                 "traffic": traffic.get("fwd0"), "traffic_source": tsrc, "avg_launch_ms": round(avg_ms, 4),
                 "launches": len(site_ms["fwd0"]), "algorithmic_flop_per_launch": flops}
            if a.precision == "bf16":
                # the bf16 conv1 forward reads the fp32 NCHW input itself and leaves the bf16 tiles behind for the
                # weight gradient: per patch 369*121*4 B in, 384*121*2 B of tiles + 64*121*2 B of output (IEEE half) out
                # = 287,012 B.  That makes it HBM-bound (its MFMA floor is ~21 us, its HBM floor ~37 us at 8 TB/s).
                # Against the strictly compulsory bytes (input in + output out, the tiles being a by-product) the
                # fraction is lower:
                # `achieved` / `frac` price SURVEY.md 8(d)'s COMPULSORY bytes only (input in + conv output out); the
                # bf16 tiles are a by-product for the kernel's own weight gradient and are reported beside it
                withby = a.batch * (BANDS * HW * HW * 4 + 384 * HW * HW * 2 + 64 * HW * HW * 2)
                without = a.batch * (BANDS * HW * HW * 4 + 64 * HW * HW * 2)
                # SURVEY.md 8(d)'s per-patch figure for this kernel is the fp32 input alone (369 * 121 * 4 B = 178,596 B ->
                # 182.9 MB per launch at B = 1024): that is what `achieved` / `frac` price since round 6 (rounds 4-5 counted
                # the 15.9 MB half output in as well: `frac_with_output`)
                nbytes = a.batch * (BANDS * HW * HW * 4)
                gbs = nbytes / (avg_ms * 1e-3) / 1e9
                r.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                          "frac": round(gbs / PEAK_HBM_GBS, 4), "algorithmic_bytes_per_launch": nbytes,
                          "frac_with_output": round(without / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                          "frac_with_byproduct": round(withby / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                          "bytes_per_launch_with_byproduct": withby,
                          "mfma_tflops": round(ach, 2), "mfma_frac": round(ach / PEAK_TFLOPS[a.precision], 4),
                          "kernel": "k_conv3x3_bf16<2,2,XN> (conv1 forward, both branches; converts the fp32 input and "
                                    "emits the bf16 tiles)"})
            roofs["fwd0"] = r
        total = a.steps * a.batch * world
        value = total / el
        per_gpu = value / world
        out = {
            "metric": "patches/sec (train step) Hang2020 369-band 11x11",
            "value": round(value, 1), "unit": "patches/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(el / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": "Hang2020 spectral+spatial attention train step (fwd + weighted CE + bwd + Adam"
                                   + (" + RCCL grad all-reduce" if world > 1 else "") + "), bands=369 11x11 classes=200",
                       "per_gpu_batch": a.batch, "global_batch": a.batch * world,
                       "parallelism": f"dp{world}", "exchange": trainer.exchange, "exchange_fallbacks": fallbacks,
                       "overlap_comm": bool(trainer.overlap), "ranks_seen": ranks_seen,
                       "rank_ms_per_step": rank_ms,      # each rank's own K steps before the closing barrier: min / max over ranks
                       "collectives_per_step": (0 if trainer.exchange in (None, "peer") else (2 if trainer.overlap else 1)),
                       "exchange_launches_per_step": (1 if trainer.exchange == "peer" else 0)},
            "library_build_id": build_id, "priming_steps_before_warmup": primed,
            "achieved_tflops_step": round(value * FLOP_PER_PATCH_STEP / 1e12, 2),
            "achieved_hbm_gbs_algorithmic": round(value * BYTES_PER_PATCH_STEP / 1e9, 1),
            "final_loss": round(final_loss, 5),
            "roofline": dict(roofs[a.site], measured_in=f"HIP events around every {max(1, a.site_stride)}th launch inside the contract's timed steps") if a.site in roofs else None,
            ("roofline_mfma" if a.site == "fwd0" else "roofline_hbm"):
                dict(roofs[other], measured_in=f"HIP events around every launch of {n_other} further steps of the same run")
                if other in roofs else None,
            "step_roofline": {
                "bound": "mfma", "per_gpu": True,
                "achieved": round(per_gpu * FLOP_PER_PATCH_STEP / 1e12, 2), "peak": PEAK_TFLOPS[a.precision],
                "unit": "TFLOP/s", "frac": round(per_gpu * FLOP_PER_PATCH_STEP / 1e12 / PEAK_TFLOPS[a.precision], 4),
                "algorithmic_flop_per_patch": FLOP_PER_PATCH_STEP,
                "hbm_gbs_algorithmic": round(per_gpu * BYTES_PER_PATCH_STEP / 1e9, 1),
                "hbm_frac_algorithmic": round(per_gpu * BYTES_PER_PATCH_STEP / 1e9 / PEAK_HBM_GBS, 4),
                "traffic": traffic.get("step_total"), "traffic_source": tsrc},
            "steady_state": steady,
            "tile_input": tile_in,
        }
        if world == 1 and not dist_on and not a.no_side:
            # side workloads, driver-timed in the same run AFTER the contract's region (20 steps each): the reference's own
            # precision, BASELINE configs[4], and the unchanged-reference-step plugin path
            fused_ms = steady["median_ms_per_step"] if steady else el / a.steps * 1e3
            trainer.close()
            del trainer, xs, ys
            torch.cuda.empty_cache()
            for name, fn in (("fp32", lambda: side_fp32(a, dev)),
                             ("ensemble24", lambda: main_ensemble24(argparse.Namespace(**dict(vars(a), steps=20, warmup=5, prime_seconds=0.0, batch=1024, site="fwd0")), emit=False)),
                             ("metadata", lambda: side_metadata(a, dev)),
                             ("multistage", lambda: side_multistage(a, dev)),
                             ("module_path", lambda: side_module_path(a, dev, fused_ms))):
                try:
                    out[name] = fn()
                except Exception as e:      # noqa: BLE001 -- a side number must never cost the headline line
                    out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            trainer = None
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_batch, a.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist_on:
        if trainer is not None:
            trainer.close()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
