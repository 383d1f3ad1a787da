R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2c; mkdir -p $O
cd $R
./tools/bin/probe_launch > $O/probe_launch.log 2>&1
python bench.py --no-cpu-baseline --steps 100 > $O/bench_new.json 2> $O/bench_new.err
DTA_BN_FINALIZE_LAUNCH=1 python bench.py --no-cpu-baseline --steps 100 > $O/bench_finlaunch.json 2> $O/bench_finlaunch.err
python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1
tail -n 8 $O/pytest_all.log; grep "bwd_finalize\|adds" $O/probe_launch.log; cat $O/bench_new.json $O/bench_finlaunch.json | cut -c1-260
