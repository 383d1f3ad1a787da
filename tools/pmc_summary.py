"""Summarise a rocprofv3 --pmc rocpd database: per kernel, mean counter value per dispatch."""
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
rows = c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection").fetchall() \
    if "kernel_name" in cols else []
if not rows:
    print("columns:", cols)
    sys.exit(0)
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for k, cn, v, d in rows:
    if pat and pat not in k:
        continue
    acc[k][cn] += v
    disp[k].add(d)
for k in sorted(acc, key=lambda k: -len(disp[k])):
    n = len(disp[k])
    print(f"{k[:100]}  dispatches={n}")
    for cn in sorted(acc[k]):
        print(f"    {cn:<28} {acc[k][cn] / n:>16.1f}")
