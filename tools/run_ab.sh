# same-box A/B of two developer switches: usage: run_ab.sh <outdir> ["ENV=1" ...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; shift
cd $R
for rep in 1 2; do
  python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('default      ', j['ms_per_step'])" | tee -a $O/ab.log
  for e in "$@"; do
    env $e python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$e', j['ms_per_step'])" | tee -a $O/ab.log
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --steps 40 --warmup 10 > $O/kt.log 2>&1
python $R/tools/prof_summary.py $O/kt/kt_results.db 50 > $O/kt_summary.txt; head -36 $O/kt_summary.txt
