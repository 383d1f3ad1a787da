mkdir -p gpurun_out/p5
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 1000 > gpurun_out/p5/a$i.json 2>gpurun_out/p5/a.err
DTA_NO_STAGGER=1 python bench.py --no-cpu-baseline --steps 1000 > gpurun_out/p5/b$i.json 2>gpurun_out/p5/b.err
done
python bench.py --no-cpu-baseline --workload ensemble24 > gpurun_out/p5/e1.json 2>gpurun_out/p5/e.err
DTA_NO_STAGGER=1 python bench.py --no-cpu-baseline --workload ensemble24 > gpurun_out/p5/f1.json 2>gpurun_out/p5/e.err
python - <<'PY'
import json
for n in ("a1","b1","a2","b2","e1","f1"):
    d=json.loads(open(f"gpurun_out/p5/{n}.json").read().strip().splitlines()[-1]); print(n, d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
PY
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/p5/tests.txt 2>&1
tail -2 gpurun_out/p5/tests.txt
