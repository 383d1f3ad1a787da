# SQ counter pass over the bench step: usage run_sq.sh <outdir>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --steady-steps 0 --tile-steps 0 --other-steps 0"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d $O/sq -o s -- $B --steps 3 --warmup 2 > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU --kernel-trace -d $O/sq2 -o s -- $B --steps 3 --warmup 2 > $O/sq2.log 2>&1
python - <<PY
import sqlite3
from collections import defaultdict
for db in ("$O/sq/s_results.db","$O/sq2/s_results.db"):
    try:
        c=sqlite3.connect(db)
        acc=defaultdict(lambda: defaultdict(float)); n=defaultdict(set)
        for k,cn,v,d in c.execute("select kernel_name,counter_name,value,dispatch_id from counters_collection"):
            acc[k][cn]+=v; n[k].add(d)
        for k in acc:
            if "stage" in k or "apply" in k:
                print(k[:70], {cn: round(s/len(n[k])) for cn,s in acc[k].items()})
    except Exception as e: print(db, e)
PY
