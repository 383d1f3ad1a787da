# Developer tool: same-box alternation of two library builds on the bench step (no tracer):
#   <build base>; cp deeptreeattention_amd/libdta_hip.so deeptreeattention_amd/libdta_base.so; <edit, rebuild>; gpurun -- bash tools/ab_lib_step.sh
R=$GRAFT_REPO_ROOT; D=$R/deeptreeattention_amd; cd $R
cp $D/libdta_hip.so $D/libdta_new.so
for rep in 1 2 3 4; do for V in base new; do
  cp $D/libdta_$V.so $D/libdta_hip.so
  python bench.py --steps 200 --warmup 20 --no-side --no-cpu-baseline --tile-steps 0 --other-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('[$V]', 'ms_per_step', d['ms_per_step'], 'steady', d['steady_state']['median_ms_per_step'])"
done; done
cp $D/libdta_new.so $D/libdta_hip.so
