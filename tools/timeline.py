"""Developer tool: one train step's kernel timeline from a rocprofv3 --kernel-trace database (start offset, duration, gap
to the previous kernel's end, stream/queue).  usage: timeline.py <results.db> [step_index_from_end]"""
import sqlite3, sys
db = sys.argv[1]; back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(c.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "k_forward_prep" in r[0]]
i0, i1 = idx[-back - 1], idx[-back]
t0 = rows[i0][1]; prev_end = t0
print(f"step of {i1 - i0} kernels, {(rows[i1][1] - t0) / 1e3:.1f} us from prep to prep")
for r in rows[i0:i1]:
    n = r[0].replace("void ", "").replace("dta::", "")[:64]
    print(f"{(r[1] - t0) / 1e3:8.1f} us  +{(r[2] - r[1]) / 1e3:6.1f}  gap {(r[1] - prev_end) / 1e3:6.1f}  q={r[3] if qcol else '-'}  {n}")
    prev_end = max(prev_end, r[2])
