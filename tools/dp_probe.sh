# Fixed cost of each gradient exchange with ONE rank (a one-rank RCCL process group under DTA_FORCE_COLLECTIVES=1 still issues
# the step's exchange), for all three trainers: plain single process vs peer (overlapped / one launch) vs direct RCCL
# (side-stream two buckets / one collective) vs torch.distributed buckets; same box, same process per trainer.
cd $GRAFT_REPO_ROOT
python tools/dp_one_rank.py hang metadata ensemble24 2>/dev/null
