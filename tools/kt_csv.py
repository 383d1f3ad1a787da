"""Dump a rocprofv3 rocpd database to CSV so that the summaries under profiles/ can be re-derived by a reader:
    python tools/kt_csv.py KT.db out_dispatches.csv            (kernel trace: one row per dispatch: kernel, start_ns, end_ns)
    python tools/kt_csv.py --pmc PMC.db out_counters.csv       (PMC pass: one row per dispatch and counter)
Kernel names are shortened (template arguments kept) and torch's own kernels are kept too."""
import csv
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "").replace("dta::", "")
    cut = name.find("(")
    return (name[:cut] if cut > 0 else name)[:120]


def main():
    args = sys.argv[1:]
    pmc = args and args[0] == "--pmc"
    if pmc:
        args = args[1:]
    db, out = args
    c = sqlite3.connect(db)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        if pmc:
            w.writerow(["kernel", "dispatch_id", "counter", "value"])
            for k, d, cn, v in c.execute("select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection "
                                         "group by dispatch_id, counter_name order by dispatch_id"):
                w.writerow([short(k), d, cn, v])
        else:
            w.writerow(["kernel", "start_ns", "end_ns", "duration_us"])
            t0 = None
            for k, s, e in c.execute("select name, start, end from kernels order by start"):
                t0 = s if t0 is None else t0
                w.writerow([short(k), s - t0, e - t0, round((e - s) / 1e3, 3)])


if __name__ == "__main__":
    main()
