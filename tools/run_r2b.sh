R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2b; mkdir -p $O
cd $R
./tools/bin/probe_launch > $O/probe_launch.log 2>&1
python -m pytest tests/test_hip_benched_path.py -q -s > $O/pytest_new.log 2>&1
tail -n 12 $O/pytest_new.log; cat $O/probe_launch.log
