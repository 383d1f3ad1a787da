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

    def __init__(self, world, group=None, side_stream=None, rccl=None, skip_allreduce=False):
        """rccl: a RcclDirect -> every reduction is one ncclAllReduce enqueued on the compute stream itself (nothing to
        wait for afterwards); None -> torch.distributed work objects on the backend's stream.
        skip_allreduce: development measurement switch (the phase split without the collectives: replicas DIVERGE); an
        explicit constructor argument of the trainers (`_dev_skip_allreduce=True`), never read from the environment."""
        self.world, self.group = int(world), group
        self.rccl = rccl
        self.skip_allreduce = bool(skip_allreduce)
        self.grad_scale = 1.0 / self.world
        self.collectives = 0          # all-reduces issued so far (tests: <= 2 per step)
        self._pending, self._post = [], []

    _warned_skip = False

    def _ar(self, t, side=False):
        if self.skip_allreduce:      # development: the phase split without the collectives
            if not GradSync._warned_skip:
                import warnings
                warnings.warn("skip_allreduce=True: gradient all-reduces are being DROPPED (replicas diverge); "
                              "development measurement switch, never use it in a training job")
                GradSync._warned_skip = True
            return
        if self.rccl is not None:
            if side:                 # overlapped: on RcclDirect's side stream, forked from the compute stream here
                self.rccl.all_reduce_side(t)
            else:
                self.rccl.all_reduce(t)
        else:
            self._pending.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.collectives += 1

    def reduce_early(self, flat_head, alpha_grad=None, alpha_slot=None):
        """Everything except the first conv's weight gradient (ready when backward phase 1 has been enqueued).
        alpha_grad: 0-d float64 gradient whose exchange rides in `alpha_slot`, a 1-element view of flat_head."""
        if alpha_grad is not None:
            alpha_slot.copy_(alpha_grad.reshape(1))
            self._post.append((alpha_grad, alpha_slot))
        self._ar(flat_head, side=True)

    def reduce_late(self, flat_tail):
        self._ar(flat_tail, side=True)

    def reduce_all(self, flat, alpha_grad=None, alpha_slot=None):
        """Single-bucket mode (no overlap): one collective over the whole flat gradient (RCCL direct: on the compute stream
        itself, nothing to wait for afterwards)."""
        if alpha_grad is not None:
            alpha_slot.copy_(alpha_grad.reshape(1))
            self._post.append((alpha_grad, alpha_slot))
        self._ar(flat)

    def finish(self):
        """Make the compute stream wait for the reductions before the optimizer kernel."""
        for w in self._pending:
            w.wait()
        self._pending.clear()
        if self.rccl is not None:
            self.rccl.join()
        for ag, slot in self._post:
            ag.copy_(slot.reshape(ag.shape))
        self._post.clear()

    def broadcast(self, tensors, src=0):
        for t in tensors:
            dist.broadcast(t, src, group=self.group)


class _XchgHandle:
    """Owner of one dta_xchg object: destroyed (peer mappings closed, device / uncached / pinned memory freed) when the
    last reference goes -- PeerExchange.close() after its barrier, or garbage collection of a trainer that was dropped
    without close() (local teardown only, no collective: the peers' own mappings of this rank's memory stay valid until
    they close them, HIP IPC memory is reference-counted by the driver)."""

    def __init__(self, L, h):
        self.L, self.h = L, h

    def destroy(self):
        if self.h is not None:
            self.L.dta_xchg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:      # noqa: BLE001 -- interpreter shutdown
            pass


class _DeviceBlock:
    """A device allocation owned by the library, shown to torch through __cuda_array_interface__ (zero copy)."""

    def __init__(self, ptr, n, owner):
        self._owner = owner            # the _XchgHandle: the allocation lives as long as a tensor view of it exists
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class PeerExchange:
    """Gradient exchange + optimizer step through IPC-mapped peer memory (csrc/xchg.hip, dta_xchg_*): one launch on the
    compute stream per step, no collective library.  Ranks must be processes of ONE node (xGMI / same GPU in tests).

    grad: the flat float32 gradient buffer (library-owned device memory, mapped by every peer) as a torch tensor; the
    trainer points its gradient views into it.  Handles travel once, at start-up, through torch.distributed's object
    collective of `group` (any backend)."""

    def __init__(self, n_floats, group=None, timeout_s=None, max_workgroups=None, split=0):
        """max_workgroups: grid bound of the exchange launch (default 256 = one per CU).  Processes that SHARE a GPU (the
        multi-rank tests on a one-GPU box) must keep world x max_workgroups within what stays co-resident, since every
        rank's launch waits inside the kernel for the others."""
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        self._L = L
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1         # single process: the same launch without peers
        h = C.c_void_p()
        _lib.check(L.dta_xchg_create(self.rank, self.world, int(n_floats), C.byref(h)), "dta_xchg_create")
        self._h = h
        self._owner = _XchgHandle(L, h)      # (no reference back to this object: dropping the trainer frees everything)
        if timeout_s:
            L.dta_xchg_set_timeout(h, float(timeout_s))
        if max_workgroups:
            L.dta_xchg_set_max_workgroups(h, int(max_workgroups))
        # split > 0: the buffer is exchanged as head [0, split) + tail; the head's sum over the ranks can then ride in the
        # first conv's weight-gradient launch (dta_net_backward_xchg): the exchange overlaps with the backward
        self.split = int(split)
        if self.split:
            _lib.check(L.dta_xchg_set_split(h, self.split), "dta_xchg_set_split")
        self.capacity = int(L.dta_xchg_grad_capacity(h))
        mine = C.create_string_buffer(_lib.XCHG_HANDLE_BYTES)
        _lib.check(L.dta_xchg_export(h, mine), "dta_xchg_export")
        if self.world > 1:
            gathered = [None] * self.world
            dist.all_gather_object(gathered, mine.raw, group=group)
            blob = C.create_string_buffer(b"".join(gathered), _lib.XCHG_HANDLE_BYTES * self.world)
            _lib.check(L.dta_xchg_connect(h, blob), "dta_xchg_connect")
            dist.barrier(group=group)            # every rank has mapped every peer before the first step
        self.grad = torch.as_tensor(_DeviceBlock(L.dta_xchg_grad_buffer(h), self.capacity, self._owner), device="cuda")
        assert self.grad.data_ptr() == L.dta_xchg_grad_buffer(h), "torch copied the exchange buffer instead of wrapping it"
        self.grad_scale = 1.0 / self.world
        self.steps = 0

    def allreduce(self, alpha_g=None, alpha_slot=-1):
        """grad := sum over ranks (enqueued on the current stream).  alpha_g: this rank's float64 d(alpha), which enters
        the sum through slot `alpha_slot` of the buffer."""
        from . import _lib
        _lib.check(self._L.dta_xchg_allreduce(self._h, _lib.ptr(alpha_g), int(alpha_slot) if alpha_g is not None else -1,
                                              _lib.current_stream_ptr()), "dta_xchg_allreduce")
        self.steps += 1

    def reduce_head(self, alpha_g=None, alpha_slot=-1, stream=None):
        """The head segment's sum over the ranks (reduce-scatter part) of the NEXT allreduce / adam_step as a launch of its
        own on `stream` (a torch.cuda.Stream; default: the current one): the overlapped form for plans whose first-conv
        weight gradient has no combined kernel.  The head [0, split) must be complete on that stream."""
        from . import _lib
        st = _lib.current_stream_ptr() if stream is None else stream.cuda_stream
        _lib.check(self._L.dta_xchg_reduce_head(self._h, _lib.ptr(alpha_g), int(alpha_slot) if alpha_g is not None else -1, st),
                   "dta_xchg_reduce_head")

    def adam_step(self, p, m, v, alpha, alpha_g, alpha_slot, alpha_m, alpha_v, step, lr, betas, eps, zero_grad):
        from . import _lib
        _lib.check(self._L.dta_xchg_adam_step(self._h, _lib.ptr(p), _lib.ptr(m), _lib.ptr(v), p.numel(), _lib.ptr(alpha),
                                              _lib.ptr(alpha_g), -1 if alpha is None else int(alpha_slot), _lib.ptr(alpha_m),
                                              _lib.ptr(alpha_v), int(step), lr, betas[0], betas[1], eps, self.grad_scale,
                                              1 if zero_grad else 0, _lib.current_stream_ptr()), "dta_xchg_adam_step")
        self.steps += 1

    def check(self):
        """Raise if an exchange launch timed out waiting for a peer.  A read of a pinned host word (no synchronisation):
        a launch that timed out has written it when it ends, and from then on every launch of this exchange returns at
        once without applying anything (sticky abort, csrc/xchg.hip) -- the trainers call this at the start of every
        step, so a replica stops within the steps that were already enqueued instead of training on alone."""
        if self._h is not None and self._L.dta_xchg_status(self._h) != 0:
            raise RuntimeError(self._L.dta_last_error().decode())

    def close(self):
        """Collective, MANDATORY for an orderly shutdown: every rank stops using its peers' memory (barrier) before
        anybody frees it.  Every rank must call it (the barrier deadlocks otherwise).  A trainer that is simply dropped
        still frees its own allocations and mappings when it is collected (_XchgHandle), without the barrier."""
        if self._h is not None:
            torch.cuda.synchronize()
            if self.world > 1 and dist.is_initialized():
                dist.barrier(group=self.group)
                self._L.dta_xchg_disconnect(self._h)      # unmap the peers' buffers ...
                dist.barrier(group=self.group)             # ... and free only what nobody has mapped any more
            self.grad = None
            self._owner.destroy()
            self._h = None


class RcclDirect:
    """ncclAllReduce called straight from librccl.so (ctypes) ON THE COMPUTE STREAM: no torch.distributed work object,
    no communication stream, no event hops (each of which stalled the compute stream ~12 us in the torch path).  The
    communicator is bootstrapped from a unique id that travels through torch.distributed's object broadcast."""

    NCCL_FLOAT, NCCL_SUM = 7, 0

    def __init__(self, group=None):
        import ctypes as C
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        import glob
        import os
        lib = None
        # the copy torch itself loaded comes first (one RCCL instance per process), then the ROCm installation's
        cands = sorted(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")))
        for name in cands + ["librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"]:
            try:
                lib = C.CDLL(name)
                break
            except OSError:
                continue
        if lib is None:
            raise RuntimeError("librccl.so not found")

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_ubyte * 128)]
        self._lib = lib
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        uid = UniqueId()
        if self.rank == 0:
            self._ok(lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        box = [C.string_at(C.byref(uid), 128) if self.rank == 0 else None]      # all 128 bytes (the id holds NULs)
        if self.world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        C.memmove(C.byref(uid), box[0], 128)
        comm = C.c_void_p()
        self._ok(lib.ncclCommInitRank(C.byref(comm), self.world, uid, self.rank), "ncclCommInitRank")
        self._comm = comm
        self.collectives = 0

    def _ok(self, rc, what):
        if rc != 0:
            raise RuntimeError("{}: {}".format(what, self._lib.ncclGetErrorString(rc).decode()))

    def all_reduce(self, t):
        """In-place sum of a contiguous float32 device tensor, enqueued on the current stream."""
        import ctypes as C
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
        ptr = C.c_void_p(t.data_ptr())
        self._ok(self._lib.ncclAllReduce(ptr, ptr, t.numel(), self.NCCL_FLOAT, self.NCCL_SUM, self._comm,
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)), "ncclAllReduce")
        self.collectives += 1

    # ---- north_star's literal form: "RCCL all-reduce of gradients over xGMI overlapped with backward on a side HIP
    #      stream" (reference train.py:89-98: DDP reduces its buckets while earlier layers still back-propagate).  Plain
    #      HIP streams and events (torch.cuda.Stream / Event are those), no torch.distributed work objects: the side stream
    #      is forked from the compute stream by one event per bucket, and joined by one event before the optimizer ----
    def _side_objects(self):
        if getattr(self, "_stream", None) is None:
            self._stream = torch.cuda.Stream()
            self._ev_fork = [torch.cuda.Event(), torch.cuda.Event()]
            self._ev_join = torch.cuda.Event()
            self._k, self._side_pending = 0, False
        return self._stream

    def all_reduce_side(self, t):
        """In-place sum of `t` on the side stream, ordered after everything the compute stream has been given so far; the
        compute stream carries on (the first conv's weight gradient) and meets the result in join()."""
        import ctypes as C
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
        side = self._side_objects()
        ev = self._ev_fork[self._k]
        self._k ^= 1
        ev.record(torch.cuda.current_stream())
        side.wait_event(ev)
        ptr = C.c_void_p(t.data_ptr())
        self._ok(self._lib.ncclAllReduce(ptr, ptr, t.numel(), self.NCCL_FLOAT, self.NCCL_SUM, self._comm,
                                         C.c_void_p(side.cuda_stream)), "ncclAllReduce")
        self.collectives += 1
        self._side_pending = True

    def join(self):
        """The compute stream waits for the side stream's collectives (one event)."""
        if getattr(self, "_side_pending", False):
            self._ev_join.record(self._stream)
            torch.cuda.current_stream().wait_event(self._ev_join)
            self._side_pending = False

    def close(self):
        if self._comm is not None:
            torch.cuda.synchronize()
            self._lib.ncclCommDestroy(self._comm)
            self._comm = None


def probe_peer_exchange(group=None, budget_s=90.0):
    """Collective over `group`: run the crash-isolated self-test of the peer exchange (peer_probe.py, one subprocess per
    rank on this rank's device) and agree on the outcome.  True = every rank's probe mapped its peers and summed
    correctly; a fault in a probe (e.g. no peer access between two devices) costs only that subprocess."""
    import shutil
    import subprocess
    import sys
    import tempfile
    import socket
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    hosts = [None] * world
    dist.all_gather_object(hosts, socket.gethostname(), group=group)
    if len(set(hosts)) > 1:            # peers are mapped through HIP IPC: one node only
        return False, ["ranks span {} hosts".format(len(set(hosts)))] * world
    box = [tempfile.mkdtemp(prefix="dta_peer_probe_") if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    here = os.path.dirname(os.path.abspath(__file__))
    ok = False
    try:
        proc = subprocess.run([sys.executable, os.path.join(here, "peer_probe.py"), box[0], str(rank), str(world),
                               str(torch.cuda.current_device())], timeout=budget_s, capture_output=True, text=True)
        ok = proc.returncode == 0
        why = (proc.stderr or "").strip().splitlines()[-1:] if not ok else []
    except Exception as e:         # timeout, missing interpreter, ...
        why = [repr(e)]
    votes = [None] * world
    dist.all_gather_object(votes, (ok, why), group=group)
    if rank == 0:
        shutil.rmtree(box[0], ignore_errors=True)
    return all(v[0] for v in votes), [v[1] for v in votes]


def choose_exchange(prefer=None, group=None):
    """Which gradient exchange a trainer uses for world > 1: "peer" (csrc/xchg.hip, fused with Adam), "rccl" (ncclAllReduce
    on the compute stream) or "torch" (torch.distributed work objects: the portable path, gloo in the CPU tests).
    prefer=None: peer when the crash-isolated probe passes on every rank, else rccl on an RCCL process group, else torch."""
    if prefer in ("peer", "rccl", "torch"):
        return prefer
    if prefer not in (None, "auto"):
        raise ValueError("exchange must be one of None / 'auto' / 'peer' / 'rccl' / 'torch'")
    if not torch.cuda.is_available():
        return "torch"
    ok, why = probe_peer_exchange(group)
    if ok:
        return "peer"
    if dist.get_rank(group) == 0:
        import warnings
        warnings.warn("peer gradient exchange unavailable on this node ({}); using the RCCL / torch.distributed path".format(why))
    return "rccl" if dist.get_backend(group) == "nccl" else "torch"
