# Developer probe: shader / memory clocks, socket power and temperatures sampled while the bench step runs (is the part clock- or power-limited?)
cd $GRAFT_REPO_ROOT
python bench.py --steps 60000 --warmup 20 --no-side --no-cpu-baseline --tile-steps 0 --other-steps 0 --steady-steps 0 > /tmp/b.json 2>/dev/null &
P=$!
sleep 28
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|hotspot)" | head -8 | tr '\n' ';'; echo; sleep 0.7; done
wait $P
cut -c1-200 /tmp/b.json
echo; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -3
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -2
