# Round-5 development check on one box: usage  bash tools/r5_check.sh <tag> "<pytest args>" [ab-switch]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
python -m pytest $2 -m gpu -q -p no:cacheprovider > $O/tests.txt 2>&1; echo "pytest rc=$?"; tail -12 $O/tests.txt | cut -c1-250
if [ -n "$3" ]; then bash tools/ab.sh "$3" "DTA_X=0" 3 > $O/ab.txt 2>&1; cat $O/ab.txt; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-side --steady-steps 0 --tile-steps 0 --other-steps 0 --prime-seconds 0 --steps 50 --warmup 10 > $O/kt.log 2>&1
cd $R; python tools/prof_summary.py $O/kt/kt_results.db 40 > $O/kernel_trace.txt; head -32 $O/kernel_trace.txt | cut -c1-150; rm -rf $O/kt
