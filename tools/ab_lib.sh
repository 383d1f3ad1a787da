# Developer tool: per-kernel averages (rocprofv3 kernel trace) of two library builds on the same box:
#   cp deeptreeattention_amd/libdta_hip.so deeptreeattention_amd/libdta_base.so; <edit, rebuild>; gpurun -- bash tools/ab_lib.sh 'k_name|k_other' [reps]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ablib; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
D=$R/deeptreeattention_amd; PAT="${1:-total kernel}"; REPS="${2:-2}"
cp $D/libdta_hip.so $D/libdta_new.so
B="python $R/bench.py --no-cpu-baseline --no-side --steady-steps 0 --tile-steps 0 --other-steps 0 --prime-seconds 0"
for rep in $(seq 1 $REPS); do for V in base new; do
  cp $D/libdta_$V.so $D/libdta_hip.so
  rocprofv3 --kernel-trace --stats -d $O/kt_$V -o kt -- $B --steps 40 --warmup 10 > $O/kt_$V.log 2>&1
  echo "== $V"; python $R/tools/prof_summary.py $O/kt_$V/kt_results.db 50 | grep -E "total kernel|$PAT"
  rm -rf $O/kt_$V
done; done
cp $D/libdta_new.so $D/libdta_hip.so
