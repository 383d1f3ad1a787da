cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for m in both primary none; do
DTA_BENCH_SITES=$m python bench.py --no-cpu-baseline --tile-steps 0 --steady-steps 50 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$m', j['ms_per_step'], j['steady_state']['median_ms_per_step'])"
done; done
