"""Portable counter-based PRNG (test infrastructure).

A splitmix64-style integer hash of (seed, stream, index) -> 24-bit uniform in [0,1).
Pure NumPy so that this container (where the golden vectors are made from the
imported reference) and the GPU box (where /root/reference does not exist)
regenerate bit-identical inputs and weights.  Nothing here follows reference code;
the reference draws from torch's global RNG (e.g. tests/test_Hang2020.py:10).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_u64(seed, stream, n):
    """n 64-bit hashes for counters 0..n-1 under (seed, stream)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        key = _mix(np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(stream) + np.uint64(1))
        return _mix(idx * np.uint64(0x9E3779B97F4A7C15) + key)


def uniform01(seed, stream, shape):
    """float32 i.i.d. U[0,1) with 24 random bits (exactly representable)."""
    n = int(np.prod(shape)) if len(shape) else 1
    h = hash_u64(seed, stream, n)
    return ((h >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32).reshape(shape)


def uniform(seed, stream, shape, lo, hi):
    return (lo + (hi - lo) * uniform01(seed, stream, shape).astype(np.float64)).astype(np.float32)


def randint(seed, stream, shape, high):
    n = int(np.prod(shape)) if len(shape) else 1
    h = hash_u64(seed, stream, n)
    return ((h >> np.uint64(33)) % np.uint64(high)).astype(np.int64).reshape(shape)


def stream_id(name):
    """Stable small integer for a tensor name (FNV-1a, 32 bit)."""
    h = 0x811C9DC5
    for ch in name.encode():
        h = ((h ^ ch) * 0x01000193) & 0xFFFFFFFF
    return h
