"""neighbor_sample restated: torch_sparse::neighbor_sample<replace=false, directed=true>
(torch_sparse 0.6.12/13 era, csrc/cpu/neighbor_sample_cpu.cpp), the op called at reference
samplers/uniform_loader.py:64-72.  Sequential, hash map for relabelling — deliberately the
upstream structure, not the device algorithm:

    samples = seeds; to_local = {seed: first position}
    for each hop with fan-out k, for each node w added in the previous hop (hop 0: the seeds),
    at local position i:
        deg = colptr[w+1] - colptr[w];  skip if 0
        k < 0 or k >= deg : take every neighbour in CSC order
        else              : Robert Floyd — for j in deg-k .. deg-1: r = U{0..j};
                            if r was already chosen take j instead
        for each chosen offset off: v = row[off]; new local id len(samples) if unseen;
            emit col <- i (target), row <- local(v) (source), edge <- off

Deviation that is documented, not hidden: upstream draws r with one torch::randint call per
draw from the CPU generator (and, as written upstream, from [0, j) rather than [0, j]); that
stream cannot be reproduced by any counter-based device generator, so draws here come from
Philox4x32-10 keyed by (rng_seed, hop) and counted by (draw, local target, rng_counter), with
the textbook inclusive range that makes every k-subset equiprobable.  The device sampler is
bit-exact against THIS function; against upstream the equivalence is distributional.
"""
import numpy as np

from .philox import mulhi_range, philox4x32_10


def _draws(k, deg, t_local, hop, rng_seed, rng_counter):
    """r_t = U{0 .. deg-k+t} for t in 0..k-1 (one Philox word per draw)."""
    t = np.arange(k, dtype=np.int64)
    seed_lo = np.uint32(rng_seed & 0xFFFFFFFF)
    seed_hi = np.uint32(((rng_seed >> 32) + hop) & 0xFFFFFFFF)
    ctr_lo = np.uint32(rng_counter & 0xFFFFFFFF)
    ctr_hi = np.uint32((rng_counter >> 32) & 0xFFFFFFFF)
    words = philox4x32_10((t >> 2).astype(np.uint32), np.uint32(t_local & 0xFFFFFFFF), ctr_lo, ctr_hi,
                          seed_lo, seed_hi)
    words = np.stack(words, axis=0)  # [4, k]
    u = words[t & 3, np.arange(k)]
    return mulhi_range(u, (deg - k + t + 1).astype(np.uint64))


def neighbor_sample(colptr, row, seeds, num_neighbors, rng_seed=0, rng_counter=0):
    """Returns (node, src_local, dst_local, edge_id, hop_nodes, hop_edges) as int64 arrays/lists."""
    colptr = np.asarray(colptr, dtype=np.int64)
    row = np.asarray(row, dtype=np.int64)
    samples = [int(s) for s in seeds]
    to_local = {}
    for i, v in enumerate(samples):
        to_local.setdefault(v, i)
    rows, cols, edges = [], [], []
    hop_nodes, hop_edges = [len(samples)], [0]
    begin, end = 0, len(samples)
    for hop, k in enumerate(num_neighbors):
        for i in range(begin, end):
            w = samples[i]
            cs, ce = int(colptr[w]), int(colptr[w + 1])
            deg = ce - cs
            if deg == 0:
                continue
            if k < 0 or k >= deg:
                offsets = range(cs, ce)
            else:
                r = _draws(k, deg, i, hop, rng_seed, rng_counter)
                chosen = set()
                offsets = []
                for t in range(k):
                    j = deg - k + t
                    c = int(r[t])
                    if c in chosen:
                        c = j
                    chosen.add(c)
                    offsets.append(cs + c)
            for off in offsets:
                v = int(row[off])
                if v not in to_local:
                    to_local[v] = len(samples)
                    samples.append(v)
                cols.append(i)
                rows.append(to_local[v])
                edges.append(off)
        begin, end = end, len(samples)
        hop_nodes.append(len(samples))
        hop_edges.append(len(edges))
    as64 = lambda a: np.asarray(a, dtype=np.int64)
    return as64(samples), as64(rows), as64(cols), as64(edges), hop_nodes, hop_edges


def segment_ptr(dst_local, num_nodes):
    """CSR over local targets for an edge list grouped by target (what the op emits)."""
    counts = np.bincount(np.asarray(dst_local, dtype=np.int64), minlength=num_nodes).astype(np.int64)
    ptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(counts, out=ptr[1:])
    return ptr


def random_neighbor(colptr, row, start, rng_seed=0, rng_counter=0):
    """torch_cluster.random_walk(row, col, start, walk_length=1)[:, 1] restated (reference
    samplers/utils.py:22): one uniform neighbour per start node, the node itself if it has
    none.  Philox counter (i>>2, 0x52570000 + (i>>34), rng_counter), key rng_seed."""
    colptr = np.asarray(colptr, dtype=np.int64)
    row = np.asarray(row, dtype=np.int64)
    start = np.asarray(start, dtype=np.int64)
    n = start.shape[0]
    i = np.arange(n, dtype=np.int64)
    words = philox4x32_10((i >> 2).astype(np.uint32), (np.int64(0x52570000) + (i >> 34)).astype(np.uint32),
                          np.uint32(rng_counter & 0xFFFFFFFF), np.uint32((rng_counter >> 32) & 0xFFFFFFFF),
                          np.uint32(rng_seed & 0xFFFFFFFF), np.uint32((rng_seed >> 32) & 0xFFFFFFFF))
    words = np.stack(words, axis=0)
    u = words[i & 3, np.arange(n)] if n else np.zeros(0, dtype=np.uint32)
    cs = colptr[start]
    deg = colptr[start + 1] - cs
    out = start.copy()
    has = deg > 0
    pick = mulhi_range(u[has], deg[has].astype(np.uint64))
    out[has] = row[cs[has] + pick]
    return out


def random_nodes(num_nodes, n, rng_seed=0, rng_counter=0):
    """torch.randint(0, num_nodes, (n,)) restated on the build's Philox stream (reference
    samplers/utils.py:30-31, negative samples): counter (i>>2, 0x4E450000 + (i>>34),
    rng_counter), key rng_seed, word i&3, range reduction mulhi32(u, num_nodes)."""
    i = np.arange(n, dtype=np.int64)
    words = philox4x32_10((i >> 2).astype(np.uint32), (np.int64(0x4E450000) + (i >> 34)).astype(np.uint32),
                          np.uint32(rng_counter & 0xFFFFFFFF), np.uint32((rng_counter >> 32) & 0xFFFFFFFF),
                          np.uint32(rng_seed & 0xFFFFFFFF), np.uint32((rng_seed >> 32) & 0xFFFFFFFF))
    words = np.stack(words, axis=0)
    u = words[i & 3, np.arange(n)] if n else np.zeros(0, dtype=np.uint32)
    return mulhi_range(u, np.full(n, num_nodes, dtype=np.uint64)).astype(np.int64)
