set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2a; mkdir -p $O
cd $R
python -m pytest tests/test_hip_benched_path.py -q -s > $O/pytest_new.log 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $B --steps 20 --warmup 5 > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f -- $B --steps 3 --warmup 2 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w -- $B --steps 3 --warmup 2 > $O/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $O/sq -o s -- $B --steps 3 --warmup 2 > $O/sq.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --kernel-trace -d $O/rd -o r -- $B --steps 3 --warmup 2 > $O/rd.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -d $O/wr -o r -- $B --steps 3 --warmup 2 > $O/wr.log 2>&1
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1
tail -3 $O/pytest_new.log $O/pytest_all.log; cat $O/bench.json
