"""Host-side data-parallel plumbing of the fused trainer (one process per GPU, torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests).  No kernels here.

The Hang2020 train step has exactly one exchange: the sum of the flat gradient (3.6 MB fp32, the float64 alpha's
gradient riding in one fp32 slot of it) over ranks; BatchNorm statistics stay per rank (reference train.py:89-98 does
not enable sync_batchnorm).  The flat buffer is laid out [everything else | alpha slot | first-conv weights] so that the
first all-reduce (issued after backward phase 1, on the backend's communication stream) overlaps with the first conv's
weight-gradient kernel, the last and largest piece; without overlap the whole buffer is ONE collective."""
import os

import torch
import torch.distributed as dist


def shard_seed(base, rank):
    """Seed of rank `rank`'s shard of the synthetic global batch (each rank generates/loads its own patches)."""
    return int(base) + int(rank)


def flat_layout(named_sizes, late):
    """Order parameters for the flat buffer: those for which late(name) is true go last.
    Returns (ordered [(name, numel)], split offset, total)."""
    early = [(k, n) for k, n in named_sizes if not late(k)]
    tail = [(k, n) for k, n in named_sizes if late(k)]
    split = sum(n for _, n in early)
    return early + tail, split, split + sum(n for _, n in tail)


class GradSync:
    """Gradient all-reduce (sum) of the flat fp32 buffer; averaging is folded into the optimizer kernel (grad_scale).

    Two buckets when overlapping ([everything else | first-conv weights]: the first is reduced while the first conv's
    weight gradient is still being computed), ONE collective over the whole buffer otherwise.  The collectives are
    issued asynchronously from the compute stream: the backend's own communication stream waits for the work enqueued
    so far, runs next to whatever the compute stream does afterwards, and `finish()` makes the compute stream wait for
    them -- two stream hops per collective (a private side stream around a blocking call costs four; measured with a
    one-rank RCCL group: 29 us between the last reduction kernel and the optimizer for the exposed bucket).

    The float64 alpha gradient travels inside the fp32 buffer ("fold into the fp32 buffer, keep the fp64
    master").  The fused trainer lets the backward kernels fill that slot and the optimizer kernel read it
    (dta_net_backward_dp / dta_adam_step_dp), so nothing runs around the collective; callers without those kernels pass
    (alpha_grad, alpha_slot) and get the two copies done here."""

    def __init__(self, world, group=None, side_stream=None):
        self.world, self.group = int(world), group
        self.grad_scale = 1.0 / self.world
        self.collectives = 0          # all-reduces issued so far (tests: <= 2 per step)
        self._pending, self._post = [], []

    def _ar(self, t):
        if os.environ.get("DTA_SKIP_ALLREDUCE") == "1":      # development: the phase split without the collectives
            return
        self._pending.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.collectives += 1

    def reduce_early(self, flat_head, alpha_grad=None, alpha_slot=None):
        """Everything except the first conv's weight gradient (ready when backward phase 1 has been enqueued).
        alpha_grad: 0-d float64 gradient whose exchange rides in `alpha_slot`, a 1-element view of flat_head."""
        if alpha_grad is not None:
            alpha_slot.copy_(alpha_grad.reshape(1))
            self._post.append((alpha_grad, alpha_slot))
        self._ar(flat_head)

    def reduce_late(self, flat_tail):
        self._ar(flat_tail)

    def reduce_all(self, flat, alpha_grad=None, alpha_slot=None):
        """Single-bucket mode (no overlap): one collective over the whole flat gradient."""
        self.reduce_early(flat, alpha_grad, alpha_slot)

    def finish(self):
        """Make the compute stream wait for the reductions before the optimizer kernel."""
        for w in self._pending:
            w.wait()
        self._pending.clear()
        for ag, slot in self._post:
            ag.copy_(slot.reshape(ag.shape))
        self._post.clear()

    def broadcast(self, tensors, src=0):
        for t in tensors:
            dist.broadcast(t, src, group=self.group)


def kept_anywhere(local, group=None, device="cpu"):
    """Year-ensemble step under data parallelism: `local[i]` says whether THIS rank's shard of year i is non-zero;
    returns, identically on every rank, whether ANY rank kept year i (those years are stepped everywhere; ranks
    that skipped one contribute zero gradients, as DDP does for unused parameters)."""
    flags = torch.tensor([1.0 if k else 0.0 for k in local], device=device)
    dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)
    return [f > 0 for f in flags.tolist()]
