"""Summarise a rocprofv3 --kernel-trace rocpd database: per-kernel count / total / avg / min / max (us)."""
import sqlite3
import sys

db = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c = sqlite3.connect(db)
rows = list(c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                      "max(end-start)/1e3 from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"# {db}: total kernel time {tot:.1f} us over {steps} steps = {tot / steps:.1f} us/step")
print(f"{'kernel':<78} {'calls':>6} {'total_us':>10} {'pct':>6} {'avg_us':>9} {'min_us':>9} {'max_us':>9}")
for r in rows[:45]:
    print(f"{r[0][:78]:<78} {r[1]:>6} {r[2]:>10.1f} {100 * r[2] / tot:>6.1f} {r[3]:>9.2f} {r[4]:>9.2f} {r[5]:>9.2f}")
