cd $GRAFT_REPO_ROOT
for b in 64 128 256 512 1024 2048 4096 8192; do   # (--no-side: the headline workload only)
python bench.py --no-cpu-baseline --batch $b --steps 100 --warmup 10 --steady-steps 0 --tile-steps 0 --other-steps 0 --no-side 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('B=$b', j['ms_per_step'], 'ms', int(j['value']), 'patches/s loss', j['final_loss'])"
done
