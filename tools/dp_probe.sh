# Fixed cost of each gradient exchange with ONE rank (a one-rank process group still issues the step's exchange):
# plain single process vs peer launch vs direct RCCL vs torch.distributed buckets, same box, same run.
cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', j['ms_per_step'], j['steady_state']['median_ms_per_step'])"; }
python bench.py --no-cpu-baseline --no-side --tile-steps 0 2>/dev/null | line plain
export DTA_FORCE_COLLECTIVES=1
MASTER_PORT=29571 python bench.py --no-cpu-baseline --tile-steps 0 --exchange peer 2>/dev/null | line peer
MASTER_PORT=29572 python bench.py --no-cpu-baseline --tile-steps 0 --exchange rccl 2>/dev/null | line rccl
MASTER_PORT=29573 python bench.py --no-cpu-baseline --tile-steps 0 --exchange torch 2>/dev/null | line torch_two_buckets
MASTER_PORT=29574 python bench.py --no-cpu-baseline --tile-steps 0 --exchange torch --no-overlap 2>/dev/null | line torch_one_bucket
unset DTA_FORCE_COLLECTIVES
python bench.py --no-cpu-baseline --no-side --tile-steps 0 2>/dev/null | line plain_again
