"""Crash-isolated self-test of the peer gradient exchange (csrc/xchg.hip), run as ONE SUBPROCESS PER RANK before a
trainer trusts the exchange on an unknown node:

    python deeptreeattention_amd/peer_probe.py <rendezvous dir> <rank> <world> <device ordinal>

The ranks' probes find each other through files in the rendezvous directory (IPC handles, two barriers), map each
other's buffers, run all-reduces over 256 Ki floats with known contents (written by a kernel on the exchange's own
stream immediately before it, as a train step's gradients are) -- three with the one-segment buffer, then four with the
head / tail split of the overlapped trainers, the last two with the head's reduce-scatter as a launch of its own
(the overlapped form's flags and staging area) -- and compare every result bit for bit with the sum in rank order.  Exit code 0 = the exchange works between these devices.  A GPU memory fault, a missing peer
mapping or a wrong sum ends only the probe process; the parent (dist.probe_peer_exchange) then falls back to RCCL.
No torch import here: ctypes on libdta_hip.so and libamdhip64.so only, so a probe starts in well under a second."""
import ctypes as C
import os
import sys
import time

import numpy as np

N = 1 << 18
N_REAL = 900788      # the flat gradient buffer of Hang2020(369, 200): 900,736 parameters + slots, the size a train step exchanges
BURST = 12           # (fewer when many ranks share the probe: see _burst)


def _pattern(rank, step, n):
    i = np.arange(n, dtype=np.int64)
    return (((i * 2654435761 + rank * 40503 + step * 9973) % 2001 - 1000).astype(np.float32) / 1024.0)


def _wait_files(d, prefix, world, deadline):
    while True:
        if all(os.path.exists(os.path.join(d, f"{prefix}{r}")) for r in range(world)):
            return
        if time.time() > deadline:
            raise TimeoutError(f"peer probe: ranks missing at '{prefix}'")
        time.sleep(0.01)


def _put(d, name, data=b"1"):
    tmp = os.path.join(d, name + ".tmp")
    with open(tmp, "wb") as f:
        f.write(data)
    os.rename(tmp, os.path.join(d, name))


def _phase(L, hip, d, rank, world, deadline, tag, split, n=N, burst=0):
    """One exchange object: handles through files, three (split = 0: the one-segment form the non-overlapped trainers use) or
    four all-reduces with known contents; with a split the last two run the HEAD segment's reduce-scatter as its own
    launch first (dta_xchg_reduce_head: the device code the first conv's weight-gradient launch carries in its spare
    workgroups) -- the overlapped form's flags and staging between these very devices.
    burst > 0: on the SAME exchange object (one create / map / unmap cycle more per probe made eight probes sharing a test GPU
    trip over IPC handles of freed-and-reallocated buffers), `burst` more fill -> all-reduce -> verify triples enqueued BACK
    TO BACK with no host synchronisation between steps -- every step's gradients are written by a kernel immediately before
    the exchange that peers read them in, every step's sums are checked on the device (dta_xchg_selftest_verify) before the
    next fill overwrites them: the train loop's situation, at the train step's size when n = N_REAL.  A node whose
    kernel-boundary write-back towards PEER devices is lazier than towards the same device (the one assumption of
    csrc/xchg.hip that a one-GPU box cannot exercise) fails here, and the trainers fall back to RCCL."""
    import _lib
    h = C.c_void_p()
    _lib.check(L.dta_xchg_create(rank, world, n, C.byref(h)), "dta_xchg_create")
    L.dta_xchg_set_timeout(h, 20.0 if burst else 10.0)
    L.dta_xchg_set_max_workgroups(h, 32)          # the probes of a shared-GPU test box must stay co-resident
    if split:
        _lib.check(L.dta_xchg_set_split(h, split), "dta_xchg_set_split")
    mine = C.create_string_buffer(_lib.XCHG_HANDLE_BYTES)
    _lib.check(L.dta_xchg_export(h, mine), "dta_xchg_export")
    _put(d, f"{tag}h{rank}", mine.raw)
    _wait_files(d, f"{tag}h", world, deadline)
    blob = b"".join(open(os.path.join(d, f"{tag}h{r}"), "rb").read() for r in range(world))
    _lib.check(L.dta_xchg_connect(h, C.create_string_buffer(blob, len(blob))), "dta_xchg_connect")
    _put(d, f"{tag}c{rank}")
    _wait_files(d, f"{tag}c", world, deadline)
    g = L.dta_xchg_grad_buffer(h)
    out = np.empty(n, np.float32)
    ok = True
    for step in range(4 if split else 3):
        # the buffer is written BY A KERNEL on the exchange's own stream and the exchange follows it with no host
        # synchronisation in between: exactly the train step's situation (csrc/xchg.hip relies on the kernel-boundary
        # write-back making the gradients visible to the peers' system-scope loads)
        _lib.check(L.dta_xchg_selftest_fill(h, step, None), "dta_xchg_selftest_fill")
        if split and step >= 2:
            _lib.check(L.dta_xchg_reduce_head(h, None, -1, None), "dta_xchg_reduce_head")
        _lib.check(L.dta_xchg_allreduce(h, None, -1, None), "dta_xchg_allreduce")
        if hip.hipDeviceSynchronize() != 0:
            raise RuntimeError("exchange kernel failed")
        if L.dta_xchg_status(h) != 0:
            raise RuntimeError(L.dta_last_error().decode())
        if hip.hipMemcpy(out.ctypes.data, g, 4 * n, 2) != 0:
            raise RuntimeError("hipMemcpy D2H failed")
        want = _pattern(0, step, n)
        for r in range(1, world):
            want = want + _pattern(r, step, n)
        ok = ok and np.array_equal(out, want)
    if burst:
        for step in range(burst):
            _lib.check(L.dta_xchg_selftest_fill(h, 100 + step, None), "dta_xchg_selftest_fill")
            if split and (step & 1):
                _lib.check(L.dta_xchg_reduce_head(h, None, -1, None), "dta_xchg_reduce_head")
            _lib.check(L.dta_xchg_allreduce(h, None, -1, None), "dta_xchg_allreduce")
            _lib.check(L.dta_xchg_selftest_verify(h, 100 + step, None), "dta_xchg_selftest_verify")
        if hip.hipDeviceSynchronize() != 0:
            raise RuntimeError("exchange kernel failed")
        if L.dta_xchg_status(h) != 0:
            raise RuntimeError(L.dta_last_error().decode())
        bad = L.dta_xchg_selftest_mismatches(h)
        if bad:
            sys.stderr.write(f"peer probe: {bad} wrong elements in {burst} back-to-back exchanges of {n} floats\n")
        ok = ok and bad == 0
    _put(d, f"{tag}d{rank}")
    _wait_files(d, f"{tag}d", world, deadline)        # nobody unmaps while a peer may still read
    L.dta_xchg_disconnect(h)
    _put(d, f"{tag}e{rank}")
    _wait_files(d, f"{tag}e", world, deadline)        # ... and nobody frees what a peer still has mapped
    L.dta_xchg_destroy(h)
    return ok


def main(d, rank, world, device, budget_s=80.0):
    import _lib          # the package's ctypes binding, imported by path: the package itself would pull in torch
    deadline = time.time() + budget_s
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    if hip.hipSetDevice(int(device)) != 0:
        raise RuntimeError("hipSetDevice failed")
    L = _lib.lib()
    ok = _phase(L, hip, d, rank, world, deadline, "a", 0)
    # the two-segment buffer of the overlapped trainers: head = 3/4 of the buffer (a multiple of 4 floats), plain and overlapped
    # ... on the REAL layout (900,788 floats: Hang2020(369, 200)'s flat gradient buffer), followed on the same object by a burst
    # of back-to-back steps whose sums are checked on the device.  (Ranks of a development box SHARE one GPU and are time-sliced
    # against each other's spinning exchange launches: the burst shrinks with the world size there.)
    nburst = BURST if world <= 2 else max(4, BURST * 2 // world)
    ok = _phase(L, hip, d, rank, world, deadline, "b", (3 * N_REAL // 4) & ~3, n=N_REAL, burst=nburst) and ok
    return 0 if ok else 3


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.exit(main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])))
