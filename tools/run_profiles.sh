# Round artifacts on one box: usage run_profiles.sh <tag>   (writes gpurun_out/<tag>/...; tools/copy_profiles.sh <tag> rNN
# copies them into profiles/).  Raw rocprofv3 databases are dumped to CSV (tools/kt_csv.py) before they are removed, so every
# summary can be re-derived from files that are kept.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --site wgrad0 --no-cpu-baseline --no-side > $O/bench_wgrad0.json 2>> $O/bench_default.err
python bench.py --workload ensemble24 --steps 100 --warmup 10 > $O/bench_ensemble24.json 2>> $O/bench_default.err
python bench.py --workload ensemble24 --steps 100 --warmup 10 --site wgrad0 > $O/bench_ensemble24_wgrad0.json 2>> $O/bench_default.err
bash tools/dp_probe.sh > $O/dp_one_rank.txt 2>> $O/bench_default.err
# batch sweep of the fused step ({256 ... 8192} on one GPU)
bash tools/batch_sweep.sh > $O/batch_sweep.txt 2>> $O/bench_default.err
# same-box alternation of this round's structural change (developer library: the switch does not exist in the product one)
python tools/multistagebench.py 128 40 > $O/multistage.txt 2>> $O/bench_default.err
python tools/multistagebench.py 1024 40 >> $O/multistage.txt 2>> $O/bench_default.err
./tools/bin/probe_stream > $O/probe_stream.txt 2>&1
./tools/bin/probe_events > $O/probe_events.txt 2>&1
python tools/inferbench.py > $O/infer.txt 2>> $O/bench_default.err
python tools/ms_predict_bench.py > $O/multistage_predict.txt 2>> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-side --steady-steps 0 --tile-steps 0 --other-steps 0 --prime-seconds 0"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $B --steps 50 --warmup 10 > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f -- $B --steps 3 --warmup 2 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w -- $B --steps 3 --warmup 2 > $O/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $O/sq -o s -- $B --steps 3 --warmup 2 > $O/sq.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt_e24 -o kt -- python $R/bench.py --workload ensemble24 --steps 30 --warmup 5 --prime-seconds 0 > $O/kt_e24.log 2>&1
E24="python $R/bench.py --workload ensemble24 --prime-seconds 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_e24 -o f -- $E24 --steps 3 --warmup 2 > $O/fetch_e24.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write_e24 -o w -- $E24 --steps 3 --warmup 2 > $O/write_e24.log 2>&1
cd $R
python tools/prof_summary.py $O/kt_e24/kt_results.db 35 > $O/kernel_trace_ensemble24.txt
cd /tmp; rocprofv3 --kernel-trace --stats -d $O/kt_ms -o kt -- python $R/tools/ms_trace.py 128 20 > $O/kt_ms.log 2>&1; cd $R
python tools/prof_summary.py $O/kt_ms/kt_results.db 20 > $O/kernel_trace_multistage_B128.txt; rm -rf $O/kt_ms
# ensemble24: 3 years x 256 crops x 2 reads of the fp32 crop (forward + conv1 weight gradient) + parameter state of 3 networks
python tools/step_traffic.py --workload ensemble24 --batch 256 --algo-bytes $((3*256*2*369*576*4 + 3*4*4*521432)) --trace $O/kt_e24/kt_results.db --fetch $O/fetch_e24/f_results.db --write $O/write_e24/w_results.db --trace-steps 35 --pmc-steps 5 --out $O/traffic_ensemble24.json > $O/traffic_ensemble24.txt
python tools/prof_summary.py $O/kt/kt_results.db 60 > $O/kernel_trace.txt
python tools/step_traffic.py --trace $O/kt/kt_results.db --fetch $O/fetch/f_results.db --write $O/write/w_results.db --sq $O/sq/s_results.db --trace-steps 60 --pmc-steps 5 --out $O/traffic_step.json > $O/traffic_step.txt
# the raw per-dispatch data, kept (a few hundred KB): kernel trace rows and PMC counter rows
python tools/kt_csv.py $O/kt/kt_results.db $O/kernel_trace_dispatches.csv
python tools/kt_csv.py $O/kt_e24/kt_results.db $O/kernel_trace_ensemble24_dispatches.csv
python tools/kt_csv.py --pmc $O/fetch/f_results.db $O/pmc_fetch.csv
python tools/kt_csv.py --pmc $O/write/w_results.db $O/pmc_write.csv
python tools/kt_csv.py --pmc $O/sq/s_results.db $O/pmc_sq.csv
# scratch / register check of every kernel of both traces (private segment must be zero)
python - > $O/private_segment.txt 2>&1 <<PY
import sqlite3
for tag, db in (("hang2020 B=1024 bf16", "$O/kt/kt_results.db"), ("ensemble24", "$O/kt_e24/kt_results.db")):
    c = sqlite3.connect(db)
    print("==", tag)
    try:
        cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
        seg = [x for x in cols if "private" in x.lower() or "scratch" in x.lower()]
        print("columns:", seg)
        for r in c.execute("select name, max(%s) from kernels group by name" % seg[0]):
            print(r[1], r[0][:100])
    except Exception as e:
        print("no per-dispatch private-segment column in this trace:", e)
PY
rm -rf $O/kt $O/fetch $O/write $O/sq $O/kt_e24 $O/fetch_e24 $O/write_e24
head -5 $O/kernel_trace.txt; head -12 $O/traffic_step.txt; cut -c1-400 $O/bench_default.json
