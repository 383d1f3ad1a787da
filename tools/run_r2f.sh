R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2f; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1
tail -n 25 $O/pytest_all.log
python bench.py --no-cpu-baseline --steps 50 > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-1500
