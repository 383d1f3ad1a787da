"""Host-side data-parallel plumbing of the fused trainer (one process per GPU, torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests).  No kernels here.

The Hang2020 train step has exactly one exchange: the sum of the flat gradient (3.6 MB fp32 + the float64 alpha)
over ranks; BatchNorm statistics stay per rank (reference train.py:89-98 does not enable sync_batchnorm).  The
flat buffer is laid out [everything else | first-conv weights] so that the first all-reduce (issued after backward
phase 1, on a side HIP stream) overlaps with the first conv's weight-gradient kernel, the last and largest piece."""
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
    """Two-phase gradient all-reduce (sum); averaging is folded into the optimizer kernel (grad_scale)."""

    def __init__(self, world, group=None, side_stream=None):
        self.world, self.group, self.side = int(world), group, side_stream
        self.grad_scale = 1.0 / self.world

    def _ar(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def _on_side(self, tensors):
        if self.side is None:
            for t in tensors:
                self._ar(t)
            return
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            for t in tensors:
                self._ar(t)

    def reduce_early(self, flat_head, alpha_grad=None):
        """Everything except the first conv's weight gradient (ready when backward phase 1 has been enqueued)."""
        self._on_side([flat_head] + ([alpha_grad] if alpha_grad is not None else []))

    def reduce_late(self, flat_tail):
        self._on_side([flat_tail])

    def finish(self):
        """Make the compute stream wait for the reductions before the optimizer kernel."""
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

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
