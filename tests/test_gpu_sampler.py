"""GPU parity: CSC build and fan-out sampler, bit-exact against the CPU oracle
(integer/index work: the bar is equality, not a tolerance)."""
import numpy as np
import pytest
import torch

from oracle import csc, sampler
from tests.helpers import random_graph

pytestmark = pytest.mark.gpu


def _csc_dev(ei, n, dev):
    from graphsage_amd import ops
    return ops.csc_build(torch.from_numpy(ei).to(dev), n)


@pytest.mark.parametrize("n,e,seed", [(50, 400, 0), (1000, 30000, 1), (7, 0, 2), (300, 5000, 3)])
def test_csc_build_bit_exact(dev, n, e, seed):
    ei = random_graph(n, e, seed) if e else np.zeros((2, 0), dtype=np.int64)
    if e:  # add duplicates so the stable order matters
        ei = np.concatenate([ei, ei[:, : e // 10]], axis=1)
    colptr, row, perm = _csc_dev(ei, n, dev)
    c0, r0, p0 = csc.to_csc(ei, n)
    assert colptr.cpu().numpy().tolist() == c0.tolist()
    assert row.cpu().numpy().tolist() == r0.tolist()
    assert perm.cpu().numpy().tolist() == p0.tolist()


def _sample_dev(colptr, row, seeds, fanout, seed, counter, dev, table=None, stamp=1):
    from graphsage_amd import ops
    n = colptr.numel() - 1
    B = len(seeds)
    node_cap, edge_cap, frontier = B, 0, B
    for k in fanout:
        he = frontier * k if k >= 0 else row.numel() * 2
        edge_cap += he
        frontier = min(he, n)
        node_cap += frontier
    buf = ops.SampleBuffers(n, node_cap, max(edge_cap, 1), dev)
    if table is None:
        table = ops.new_relabel_table(n, dev)
    ops.sample_fanout(colptr, row, torch.as_tensor(seeds, dtype=torch.int64, device=dev), fanout,
                      seed, counter, table, stamp, buf)
    st = buf.state.tolist()
    nn, ne = st[ops.ST_NODES], st[ops.ST_EDGES]
    return (buf.node[:nn].cpu().numpy(), buf.src[:ne].cpu().numpy(), buf.dst[:ne].cpu().numpy(),
            buf.eid[:ne].cpu().numpy(), buf.seg_ptr[:nn + 1].cpu().numpy(), st, table)


CASES = [
    # n, e, isolated, hubs, B, fanout, seed, counter
    (200, 3000, 5, 0, 16, [5, 3], 0, 0),
    (200, 3000, 5, 0, 16, [25, 10], 1, 7),          # k >= deg for most nodes: take-all path
    (2000, 120000, 10, 8, 64, [25, 10], 2, 123456789012),
    (2000, 120000, 10, 8, 33, [70, 5], 3, 1),      # fan-out > 64: LDS path of the sampler
    (500, 20000, 3, 2, 20, [-1, 2], 4, 2),          # -1 = all neighbours
    (500, 20000, 3, 2, 20, [3, 3, 3], 5, 3),        # three hops
    (500, 20000, 3, 2, 1, [10], 6, 4),              # one seed, one hop
    (300, 60000, 0, 1, 8, [200, 2], 7, 5),          # big fan-out below degree
]


@pytest.mark.parametrize("n,e,iso,hubs,B,fanout,seed,counter", CASES)
def test_sampler_bit_exact(dev, n, e, iso, hubs, B, fanout, seed, counter):
    ei = random_graph(n, e, seed, isolated=iso, hubs=hubs)
    colptr, row, _ = _csc_dev(ei, n, dev)
    rng = np.random.default_rng(seed + 100)
    seeds = rng.choice(n, size=B, replace=False)
    seeds[0] = 0 if iso else seeds[0]           # an isolated seed when the graph has one
    node, src, dst, eid, seg_ptr, st, _ = _sample_dev(colptr, row, seeds, fanout, seed, counter, dev)
    c0, r0, _ = csc.to_csc(ei, n)
    onode, osrc, odst, oeid, hop_nodes, hop_edges = sampler.neighbor_sample(c0, r0, seeds, fanout, seed, counter)
    assert st[4] == 0
    assert node.tolist() == onode.tolist()
    assert eid.tolist() == oeid.tolist()
    assert dst.tolist() == odst.tolist()
    assert src.tolist() == osrc.tolist()
    assert seg_ptr.tolist() == sampler.segment_ptr(odst, len(onode)).tolist()
    hops = len(fanout)
    assert st[8:8 + hops + 1] == hop_nodes and st[24:24 + hops + 1] == hop_edges


def test_sampler_duplicate_seeds_and_table_reuse(dev):
    ei = random_graph(400, 8000, 11, isolated=4)
    colptr, row, _ = _csc_dev(ei, 400, dev)
    c0, r0, _ = csc.to_csc(ei, 400)
    table = None
    for stamp, (seeds, counter) in enumerate([([7, 7, 9, 0, 7], 0), ([1, 2, 3], 1), ([9, 8, 7, 7], 2)], start=1):
        node, src, dst, eid, seg_ptr, st, table = _sample_dev(colptr, row, seeds, [4, 4], 5, counter, dev,
                                                              table=table, stamp=stamp)
        onode, osrc, odst, oeid, _, _ = sampler.neighbor_sample(c0, r0, seeds, [4, 4], 5, counter)
        assert node.tolist() == onode.tolist() and src.tolist() == osrc.tolist()
        assert dst.tolist() == odst.tolist() and eid.tolist() == oeid.tolist()


def test_sampler_reports_overflow_and_bad_seed(dev):
    from graphsage_amd import ops
    ei = random_graph(100, 3000, 2)
    colptr, row, _ = _csc_dev(ei, 100, dev)
    buf = ops.SampleBuffers(100, 8, 6, dev)          # far too small for 4 seeds x 5
    table = ops.new_relabel_table(100, dev)
    ops.sample_fanout(colptr, row, torch.tensor([1, 2, 3, 4], device=dev), [5], 0, 0, table, 1, buf)
    st = buf.state.tolist()
    assert st[ops.ST_ERROR] & 3 and st[ops.ST_EDGES] <= 6 and st[ops.ST_NODES] <= 8
    buf2 = ops.SampleBuffers(100, 64, 64, dev)
    ops.sample_fanout(colptr, row, torch.tensor([1, 100], device=dev), [5], 0, 0, table, 2, buf2)
    assert buf2.state.tolist()[ops.ST_ERROR] & 4
    with pytest.raises(RuntimeError):
        ops.sample_fanout(colptr, row, torch.tensor([1], device=dev), [4000], 0, 0, table, 3, buf2)


def test_random_neighbor_bit_exact(dev):
    from graphsage_amd import ops
    ei = random_graph(300, 4000, 9, isolated=6)
    colptr, row, _ = _csc_dev(ei, 300, dev)
    c0, r0, _ = csc.to_csc(ei, 300)
    start = np.random.default_rng(0).integers(0, 300, 1000)
    out = ops.random_neighbor(colptr, row, torch.from_numpy(start).to(dev), 3, 77)
    assert out.cpu().numpy().tolist() == sampler.random_neighbor(c0, r0, start, 3, 77).tolist()


def test_sampler_reddit_degree_structural(dev):
    """Full-size fan-out (B=512, 25/10) on a dense graph: structural invariants instead of
    the slow Python oracle."""
    n, e = 20000, 4_000_000
    g = torch.Generator(device=dev).manual_seed(0)
    ei = torch.randint(0, n, (2, e), generator=g, device=dev)
    from graphsage_amd import ops
    colptr, row, _ = ops.csc_build(ei, n)
    B = 512
    seeds = torch.randperm(n, generator=torch.Generator().manual_seed(1))[:B].to(dev)
    buf = ops.SampleBuffers(n, B * (1 + 25 + 250), B * (25 + 250), dev)
    table = ops.new_relabel_table(n, dev)
    ops.sample_fanout(colptr, row, seeds, [25, 10], 0, 0, table, 1, buf)
    st = buf.state.tolist()
    nn, ne = st[0], st[1]
    assert st[4] == 0
    node, src, dst, eid = buf.node[:nn], buf.src[:ne], buf.dst[:ne], buf.eid[:ne]
    seg = buf.seg_ptr[:nn + 1]
    assert torch.equal(node[:B], seeds)
    assert node.unique().numel() == nn                                # first-seen relabel: unique
    assert torch.equal(row[eid], node[src])                           # edge -> source consistent
    owner = torch.searchsorted(colptr, eid, right=True) - 1           # column that owns the offset
    assert torch.equal(owner, node[dst])
    deg = (colptr[1:] - colptr[:-1])[node]
    cnt = seg[1:] - seg[:-1]
    t1 = st[9]
    k = torch.where(torch.arange(nn, device=dev) < B, 25, 10)
    want = torch.minimum(deg, k)
    want[t1:] = 0                                                    # last hop's nodes are not expanded
    assert torch.equal(cnt, want)
    assert bool((dst[1:] >= dst[:-1]).all())                         # grouped by target, ascending
    # no duplicate offsets inside a segment (sampling without replacement)
    key = dst * (colptr[-1] + 1) + eid
    assert key.unique().numel() == ne
    # first-seen order: a node's local id is <= the first edge position that references it (+seeds)
    first_pos = torch.full((nn,), ne + 1, device=dev, dtype=torch.int64).scatter_reduce(0, src, torch.arange(ne, device=dev), "amin")
    new = torch.arange(nn, device=dev) >= B
    fp = first_pos[new]
    assert bool((fp[1:] > fp[:-1]).all())


def test_random_nodes_and_bounded_random_neighbor(dev):
    from graphsage_amd import ops
    got = ops.random_nodes(12345, 3000, 5, 99, device=dev)
    assert got.cpu().numpy().tolist() == sampler.random_nodes(12345, 3000, 5, 99).tolist()
    ei = random_graph(300, 4000, 21, isolated=3)
    colptr, row, _ = _csc_dev(ei, 300, dev)
    c0, r0, _ = csc.to_csc(ei, 300)
    start = np.random.default_rng(0).integers(0, 300, 500)
    # device-side count: only the first 123 entries are walked, the rest of `out` is untouched
    out = torch.full((500,), -7, dtype=torch.int64, device=dev)
    bad = torch.from_numpy(start).to(dev)
    bad[200:] = 10 ** 12                       # ids outside the graph beyond the count: never read
    ops.random_neighbor(colptr, row, bad, 4, 6, out=out, checked=False, num_dev=torch.tensor([123], device=dev))
    want = sampler.random_neighbor(c0, r0, start[:123], 4, 6)
    assert out[:123].cpu().numpy().tolist() == want.tolist()
    assert bool((out[123:] == -7).all())
    # an id outside the graph inside the count yields 0 and touches no memory
    bad[5] = -3
    bad[6] = 300
    ops.random_neighbor(colptr, row, bad, 4, 6, out=out, checked=False, num_dev=torch.tensor([123], device=dev))
    assert int(out[5]) == 0 and int(out[6]) == 0 and int(out[7]) == int(want[7])


def test_exchange_wait_gives_up_with_an_error_bit(dev):
    """The sampler's in-launch prefix exchange waits for totals that other workgroups publish; the
    wait is bounded and ends in SAGE_ST_ERROR bit 4 instead of a grid that never drains.  Unit
    harness (sage_debug_exchange_wait): one wave over a prepared totals array in which one word's
    ready bit is never set."""
    import ctypes
    from graphsage_amd import _lib, ops
    lib = _lib.load()
    ready = 1 << 63
    vals = [ready | 5, ready | 7, 0, ready | 11] + [ready | 1] * 100       # word 2 is never published
    words = torch.tensor([v - (1 << 64) if v >= (1 << 63) else v for v in vals], dtype=torch.int64, device=dev)
    state = torch.zeros(ops.STATE_WORDS, dtype=torch.int64, device=dev)
    out = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.sage_debug_exchange_wait(ctypes.c_void_p(words.data_ptr()), len(vals), ctypes.c_void_p(state.data_ptr()),
                                      2000, ctypes.c_void_p(out.data_ptr()), stream)
    _lib.check(rc, "sage_debug_exchange_wait")
    torch.cuda.synchronize()
    assert int(state[ops.ST_ERROR]) == 16
    assert int(out) == 5 + 7 + 11 + 100                    # what was ready
    # every word ready: no error, the full sum
    words[2] = words[0]
    state.zero_()
    rc = lib.sage_debug_exchange_wait(ctypes.c_void_p(words.data_ptr()), len(vals), ctypes.c_void_p(state.data_ptr()),
                                      2000, ctypes.c_void_p(out.data_ptr()), stream)
    _lib.check(rc, "sage_debug_exchange_wait")
    assert int(state[ops.ST_ERROR]) == 0 and int(out) == 5 + 7 + 5 + 11 + 100
