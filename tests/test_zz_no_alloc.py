"""sage_sample_fanout_static (the entry point that allocates nothing: its workspace is a static
device array inside the library) against sage_sample_fanout, through the C ABI, bit-exact
(integer/index work), on two calls in a row — the second with a smaller input, so that whatever
the first call left in the shared array must not matter."""
import numpy as np
import pytest
import torch

from tests.helpers import random_graph

pytestmark = pytest.mark.gpu


def _caps(B, fanout, n):
    node_cap, edge_cap, frontier = B, 0, B
    for k in fanout:
        edge_cap += frontier * k
        frontier = min(frontier * k, n)
        node_cap += frontier
    return node_cap, edge_cap


def _outputs(buf):
    from graphsage_amd import ops
    st = buf.state.tolist()
    nn, ne = st[ops.ST_NODES], st[ops.ST_EDGES]
    return (st, buf.node[:nn].tolist(), buf.src[:ne].tolist(), buf.dst[:ne].tolist(),
            buf.eid[:ne].tolist(), buf.seg_ptr[:nn + 1].tolist())


def test_static_scratch_sampler_matches_workspace_sampler(dev):
    from graphsage_amd import ops
    n = 3000
    ei = random_graph(n, 150000, 41, isolated=7, hubs=5)
    colptr, row, _ = ops.csc_build(torch.from_numpy(ei).to(dev), n)
    rng = np.random.default_rng(3)
    # first call: B = 256 at 25/10 (the big workspace layout); second: a smaller batch and fan-out
    calls = [(256, [25, 10], 11), (40, [70, 3], 12)]
    table_a = ops.new_relabel_table(n, dev)
    table_b = ops.new_relabel_table(n, dev)
    for stamp, (B, fanout, counter) in enumerate(calls, start=1):
        seeds = torch.from_numpy(rng.choice(n, size=B, replace=False)).to(dev)
        seeds[0] = 0                                   # an isolated seed
        node_cap, edge_cap = _caps(B, fanout, n)
        want = ops.SampleBuffers(n, node_cap, edge_cap, dev)
        ops.sample_fanout(colptr, row, seeds, fanout, 9, counter, table_a, stamp, want)
        got = ops.SampleBuffers(n, node_cap, edge_cap, dev)
        got.ws = None                                  # the static entry point must not need it
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        ops.sample_fanout_static(colptr, row, seeds, fanout, 9, counter, table_b, stamp, got)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0   # no device memory taken or returned
        w, g = _outputs(want), _outputs(got)
        assert w[0][ops.ST_ERROR] == 0 and g[0][ops.ST_ERROR] == 0
        assert w[0][ops.ST_EDGES] > 0
        for a, b in zip(w, g):
            assert a == b


def test_static_scratch_sampler_rejects_oversize(dev):
    from graphsage_amd import ops
    n = 100
    ei = random_graph(n, 2000, 1)
    colptr, row, _ = ops.csc_build(torch.from_numpy(ei).to(dev), n)
    table = ops.new_relabel_table(n, dev)
    big = ops.SampleBuffers(n, 4_000_000, 4_000_000, dev)     # 64 MB of workspace: beyond the array
    with pytest.raises(RuntimeError, match="scratch"):
        ops.sample_fanout_static(colptr, row, torch.arange(4, device=dev), [5], 0, 0, table, 1, big)
