"""ctypes wrapper of oracle/sampler_ref.c (test infrastructure / CPU baseline only)."""
import ctypes
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "liboracle_sampler.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(_PATH)
        P = ctypes.c_void_p
        lib.oracle_neighbor_sample.restype = ctypes.c_int
        lib.oracle_neighbor_sample.argtypes = [P, P, P, ctypes.c_int, P, ctypes.c_int, ctypes.c_uint64,
                                               ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, P, P, P, P, P]
        _lib = lib
    return _lib


def neighbor_sample(colptr, row, seeds, num_neighbors, rng_seed=0, rng_counter=0):
    """Same contract as oracle.sampler.neighbor_sample (numpy int64 arrays in and out)."""
    lib = _load()
    colptr = np.ascontiguousarray(colptr, dtype=np.int64)
    row = np.ascontiguousarray(row, dtype=np.int64)
    seeds = np.ascontiguousarray(seeds, dtype=np.int64)
    fan = np.ascontiguousarray(num_neighbors, dtype=np.int32)
    B, n, e_total = len(seeds), len(colptr) - 1, len(row)
    frontier, node_cap, edge_cap = B, B, 0
    for k in num_neighbors:
        he = frontier * k if k >= 0 else e_total
        edge_cap += he
        frontier = min(he, n)
        node_cap += frontier
    node = np.empty(max(node_cap, 1), dtype=np.int64)
    src = np.empty(max(edge_cap, 1), dtype=np.int64)
    dst = np.empty(max(edge_cap, 1), dtype=np.int64)
    eid = np.empty(max(edge_cap, 1), dtype=np.int64)
    counts = np.zeros(40, dtype=np.int64)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.oracle_neighbor_sample(ptr(colptr), ptr(row), ptr(seeds), B, ptr(fan), len(fan),
                                    int(rng_seed), int(rng_counter), node_cap, edge_cap, ptr(node), ptr(src),
                                    ptr(dst), ptr(eid), ptr(counts))
    if rc != 0:
        raise RuntimeError(f"oracle_neighbor_sample failed ({rc})")
    nn, ne = int(counts[0]), int(counts[1])
    hops = len(fan)
    return (node[:nn], src[:ne], dst[:ne], eid[:ne], counts[2:2 + hops + 1].tolist(),
            counts[18:18 + hops + 1].tolist())
