cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --tile-steps 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', j['ms_per_step'], j['steady_state']['median_ms_per_step'])"; }
run plain
export DTA_FORCE_COLLECTIVES=1 MASTER_PORT=29573
run forced
export MASTER_PORT=29575 DTA_SKIP_ALLREDUCE=1
run forced_phases_only
unset DTA_SKIP_ALLREDUCE; export MASTER_PORT=29577
python bench.py --no-cpu-baseline --tile-steps 0 --no-overlap 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('forced_no_overlap', j['ms_per_step'], j['steady_state']['median_ms_per_step'])"
