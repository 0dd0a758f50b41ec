"""Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
SC'11), restated in numpy.  Pinned by the Random123 known-answer vectors in
tests/test_oracle_kat.py.  This is the counter-based generator that stands in for the
upstream sampler's torch::randint stream (torch_sparse neighbor_sample, called at reference
samplers/uniform_loader.py:64), which no device implementation can reproduce.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """All arguments broadcastable uint32 arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3, k0, k1 = np.broadcast_arrays(
        *[np.asarray(a, dtype=np.uint32) for a in (c0, c1, c2, c3, k0, k1)])
    c0, c1, c2, c3, k0, k1 = [a.copy() for a in (c0, c1, c2, c3, k0, k1)]
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & MASK).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = k0 + W0
            k1 = k1 + W1
    return c0, c1, c2, c3


def mulhi_range(u, n):
    """floor(u * n / 2^32): maps a uint32 word to [0, n)."""
    return ((np.asarray(u, dtype=np.uint64) * np.asarray(n, dtype=np.uint64)) >> np.uint64(32)).astype(np.int64)
