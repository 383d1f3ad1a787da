"""Whole-step HBM traffic / MFMA-utilisation accounting from rocprofv3 rocpd databases.

    python tools/step_traffic.py --trace KT.db --fetch F.db --write W.db [--sq SQ.db] --steps-skip 2 \
        --batch 1024 --out profiles/r02_traffic_step.json

Every database comes from its own `rocprofv3 --pmc <counters> --kernel-trace -- python bench.py ...` pass (PMC passes
serialise the launches; durations are therefore taken from the plain --kernel-trace pass).  Per kernel NAME (a step
launches some kernels several times; `launches_per_step` says how often): mean bytes per launch = 2 x FETCH_SIZE
(MI355X_MICROARCH.md: gfx950 reports half the bytes of wide coalesced reads) + WRITE_SIZE, mean duration, GB/s, and the
step totals against the algorithmic bytes of SURVEY.md 8(d).  With --sq: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256
CUs x GRBM_GUI_ACTIVE / 8 XCDs) = fraction of the matrix pipes' cycles that were busy.
"""
import argparse
import json
import sys
import os
import sqlite3
from collections import defaultdict

ALGO_BYTES_PER_PATCH = 358_000
PARAM_STATE_BYTES = 4 * 3_602_944   # parameters + gradients + two Adam moments, read/written once per step


def short(name):
    name = name.replace("void ", "").replace("dta::", "")
    for a, b in (("unsigned short", "bf16"), ("StageCfg", "Cfg")):
        name = name.replace(a, b)
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def pmc_means(db, only_ours=True):
    c = sqlite3.connect(db)
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for k, cn, v, d in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        acc[k][cn] += v
        disp[k].add(d)
    return {k: ({cn: s / len(disp[k]) for cn, s in acc[k].items()}, len(disp[k])) for k in acc}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", required=True)
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--sq")
    ap.add_argument("--trace-steps", type=int, required=True, help="train steps in the --kernel-trace run (warmup + timed)")
    ap.add_argument("--pmc-steps", type=int, required=True, help="train steps in each PMC run (warmup + timed)")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--algo-bytes", type=float, default=None, help="algorithmic bytes per step of another workload (default: SURVEY 8(d) x batch + parameter state)")
    ap.add_argument("--workload", default="hang2020")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()

    c = sqlite3.connect(a.trace)
    dur = {r[0]: (r[1], r[2]) for r in c.execute("select name, count(*), avg(end-start)/1e3 from kernels group by name")}
    fetch, write = pmc_means(a.fetch), pmc_means(a.write)
    sq = pmc_means(a.sq) if a.sq else {}
    rows = []
    tot_bytes = tot_us = 0.0
    for k, (calls, avg_us) in dur.items():
        per_step = calls / a.trace_steps
        if per_step < 0.5 or k not in fetch or k not in write:
            continue          # one-off kernels (initialisation, RNG) are not part of the step
        f_kb = fetch[k][0].get("FETCH_SIZE", 0.0)
        w_kb = write[k][0].get("WRITE_SIZE", 0.0)
        nbytes = 2.0 * f_kb * 1024 + w_kb * 1024
        row = {"kernel": short(k), "launches_per_step": round(per_step, 2), "avg_us": round(avg_us, 2),
               "fetch_kb_reported": round(f_kb, 1), "write_kb_reported": round(w_kb, 1),
               "hbm_bytes_per_launch": int(nbytes), "gb_per_s": round(nbytes / (avg_us * 1e-6) / 1e9, 1),
               "us_per_step": round(avg_us * per_step, 2), "mb_per_step": round(nbytes * per_step / 1e6, 2)}
        if k in sq:
            s = sq[k][0]
            if s.get("GRBM_GUI_ACTIVE"):
                # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (8 x kernel cycles); MFMA_BUSY over all 1024 SIMDs
                row["mfma_busy_frac"] = round(s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * s["GRBM_GUI_ACTIVE"] / 8.0), 4)
            for name in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
                         "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES"):
                if name in s:
                    row[name] = round(s[name], 1)
        rows.append(row)
        tot_bytes += nbytes * per_step
        tot_us += avg_us * per_step
    rows.sort(key=lambda r: -r["us_per_step"])
    algo = a.algo_bytes if a.algo_bytes else ALGO_BYTES_PER_PATCH * a.batch + PARAM_STATE_BYTES
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deeptreeattention_amd import _lib
    out = {"note": __doc__.split("\n\n")[2].replace("\n", " ").strip(),
           "library_build_id": _lib.lib().dta_build_id().decode(),      # bench.py quotes these counters only on this build
           "workload": a.workload, "batch": a.batch, "launches_per_step": round(sum(r["launches_per_step"] for r in rows), 1),
           "kernel_us_per_step": round(tot_us, 1), "hbm_mb_per_step": round(tot_bytes / 1e6, 1),
           "algorithmic_mb_per_step": round(algo / 1e6, 1), "traffic_over_algorithmic": round(tot_bytes / algo, 2),
           "kernels": rows}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1))
    for r in rows:
        print(f"{r['kernel'][:60]:<60} x{r['launches_per_step']:<4} {r['avg_us']:>7.1f} us {r['hbm_bytes_per_launch'] / 1e6:>8.1f} MB "
              f"{r['gb_per_s']:>7.0f} GB/s" + (f"  mfma {100 * r['mfma_busy_frac']:.1f}%" if "mfma_busy_frac" in r else ""))


if __name__ == "__main__":
    main()
