#!/bin/bash
# Developer tool: same-box alternation of two environments on the default bench (no side workloads, no CPU baseline).
#   tools/ab.sh "DTA_FANIN=3" "" [rounds]      -> prints ms_per_step / steady median per run
A="$1"; B="$2"; R="${3:-3}"
export DTA_DEV_LIB=1      # the switches live in the developer library (libdta_hip_dev.so) only
for i in $(seq 1 $R); do
  for E in "$A" "$B"; do
    env $E python bench.py --steps 200 --warmup 20 --no-side --no-cpu-baseline --tile-steps 0 --other-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('[%s]' % '$E', 'ms_per_step', d['ms_per_step'], 'steady', d['steady_state']['median_ms_per_step'], 'fwd0_us', round(d['roofline']['avg_launch_ms']*1e3,1))"
  done
done
