R=$GRAFT_REPO_ROOT; D=$R/deeptreeattention_amd; cd $R
cp $D/libdta_hip.so $D/libdta_new.so
for rep in 1 2 3; do for V in base new; do
  cp $D/libdta_$V.so $D/libdta_hip.so
  a=$(python bench.py --workload ensemble24 --steps 100 --warmup 20 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  b=$(python bench.py --precision fp32 --steps 50 --warmup 10 --no-side --no-cpu-baseline --tile-steps 0 --other-steps 0 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  c=$(python bench.py --steps 200 --warmup 20 --no-side --no-cpu-baseline --tile-steps 0 --other-steps 0 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "$V ensemble24 $a fp32 $b bf16 $c"
done; done
cp $D/libdta_new.so $D/libdta_hip.so
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
