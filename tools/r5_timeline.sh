# per-dispatch kernel timelines (CSV) of the two bench workloads: usage bash tools/r5_timeline.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-side --steady-steps 0 --tile-steps 0 --other-steps 0 --prime-seconds 0 --steps 6 --warmup 4 > $O/kt.log 2>&1
rocprofv3 --kernel-trace -d $O/kt_e24 -o kt -- python $R/bench.py --workload ensemble24 --steps 6 --warmup 4 --prime-seconds 0 > $O/kt_e24.log 2>&1
cd $R
python tools/kt_csv.py $O/kt/kt_results.db $O/dispatches_hang.csv
python tools/kt_csv.py $O/kt_e24/kt_results.db $O/dispatches_e24.csv
rm -rf $O/kt $O/kt_e24
tail -45 $O/dispatches_hang.csv | cut -c1-110; echo ----; tail -60 $O/dispatches_e24.csv | cut -c1-110
