"""Known-answer tests that pin the CPU oracle (the reference ships no tests or golden
vectors — graphsage/tests/__init__.py is empty — so these are hand-derived)."""
import numpy as np
import torch

from oracle import csc, philox, sage_ref, sampler


def test_philox_random123_kat():
    # Random123 kat_vectors, philox4x32-10
    cases = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in cases:
        got = philox.philox4x32_10(*ctr, *key)
        assert tuple(int(g) for g in got) == want


def test_mulhi_range():
    assert int(philox.mulhi_range(np.uint32(0), 10)) == 0
    assert int(philox.mulhi_range(np.uint32(0xFFFFFFFF), 10)) == 9
    assert int(philox.mulhi_range(np.uint32(0x80000000), 10)) == 5


def test_to_csc_hand_example():
    # edges (row -> col): 2->0, 1->0, 0->1, 2->1, 1->0 (duplicate), 3->3
    ei = np.array([[2, 1, 0, 2, 1, 3], [0, 0, 1, 1, 0, 3]])
    colptr, row, perm = csc.to_csc(ei, 4)
    assert colptr.tolist() == [0, 3, 5, 5, 6]
    assert row.tolist() == [1, 1, 2, 0, 2, 3]          # neighbours ascending inside a column
    assert perm.tolist() == [1, 4, 0, 2, 3, 5]         # stable for the duplicate (1->0)
    assert (ei[0][perm] == row).all()


def _chain_graph():
    # in-neighbours: 0 <- {1,2,3}; 1 <- {4}; 2 <- {4,5}; 3 <- {}; 4 <- {0}; 5 <- {}
    ei = np.array([[1, 2, 3, 4, 4, 5, 0], [0, 0, 0, 1, 2, 2, 4]])
    return csc.to_csc(ei, 6)


def test_sampler_full_neighbourhood_is_deterministic():
    colptr, row, _ = _chain_graph()
    node, src, dst, eid, hop_nodes, hop_edges = sampler.neighbor_sample(colptr, row, [0], [5, 5])
    # hop 1: seed 0 takes all of {1,2,3} in CSC order; hop 2: 1 <- 4 (new), 2 <- 4 (seen), 2 <- 5, 3 <- {}
    assert node.tolist() == [0, 1, 2, 3, 4, 5]
    assert dst.tolist() == [0, 0, 0, 1, 2, 2]
    assert src.tolist() == [1, 2, 3, 4, 4, 5]
    assert eid.tolist() == [0, 1, 2, 3, 4, 5]
    assert hop_nodes == [1, 4, 6] and hop_edges == [0, 3, 6]
    # k = -1 means all neighbours as well
    node2, src2, dst2, eid2, _, _ = sampler.neighbor_sample(colptr, row, [0], [-1, -1])
    assert node2.tolist() == node.tolist() and eid2.tolist() == eid.tolist()


def test_sampler_only_new_nodes_are_expanded_and_seeds_first():
    colptr, row, _ = _chain_graph()
    # seeds 4 and 0: hop 1: 4 <- 0 (already a seed, local 1), 0 <- 1,2,3 (new: locals 2,3,4)
    node, src, dst, eid, hop_nodes, _ = sampler.neighbor_sample(colptr, row, [4, 0], [5, 5])
    assert node[:2].tolist() == [4, 0]
    assert dst[:4].tolist() == [0, 1, 1, 1] and src[:4].tolist() == [1, 2, 3, 4]
    assert hop_nodes[1] == 5
    # hop 2 expands locals 2,3,4 (= nodes 1,2,3) only; node 4 (seed) is not expanded again
    assert set(dst[4:].tolist()) <= {2, 3, 4}
    assert node.tolist() == [4, 0, 1, 2, 3, 5]          # 1 <- 4 (seen), 2 <- 4 (seen), 2 <- 5 (new)


def test_sampler_duplicate_seeds_and_isolated():
    colptr, row, _ = _chain_graph()
    node, src, dst, eid, hop_nodes, hop_edges = sampler.neighbor_sample(colptr, row, [3, 3, 5], [2])
    assert node.tolist() == [3, 3, 5] and len(src) == 0 and hop_edges == [0, 0]
    node, src, dst, eid, _, _ = sampler.neighbor_sample(colptr, row, [1, 1], [2])
    # both copies of seed 1 are expanded; the source maps to ONE new local id
    assert node.tolist() == [1, 1, 4] and dst.tolist() == [0, 1] and src.tolist() == [2, 2]


def test_sampler_floyd_invariants_and_uniformity():
    n = 40
    ei = np.stack([np.arange(1, n), np.zeros(n - 1, dtype=np.int64)])  # node 0 has 39 in-neighbours
    colptr, row, _ = csc.to_csc(ei, n)
    k = 5
    counts = np.zeros(n - 1)
    trials = 4000
    for c in range(trials):
        node, src, dst, eid, _, _ = sampler.neighbor_sample(colptr, row, [0], [k], rng_seed=7, rng_counter=c)
        assert len(eid) == k and len(set(eid.tolist())) == k          # without replacement
        assert (eid >= colptr[0]).all() and (eid < colptr[1]).all()
        assert (row[eid] == node[src]).all() and (dst == 0).all()
        counts[eid] += 1
    expected = trials * k / (n - 1)
    chi2 = ((counts - expected) ** 2 / expected).sum()
    assert chi2 < 80.0, chi2   # 38 dof: p(chi2 > 80) ~ 1e-4
    # different counters / seeds give different draws
    a = sampler.neighbor_sample(colptr, row, [0], [k], 7, 0)[3]
    b = sampler.neighbor_sample(colptr, row, [0], [k], 8, 0)[3]
    assert a.tolist() != b.tolist()


def test_random_neighbor():
    colptr, row, _ = _chain_graph()
    out = sampler.random_neighbor(colptr, row, np.array([0, 1, 3, 2, 5]), 3, 9)
    assert out[0] in (1, 2, 3) and out[1] == 4 and out[2] == 3 and out[3] in (4, 5) and out[4] == 5


def test_scatter_kats():
    x_j = torch.tensor([[1., -2.], [3., -4.], [-5., -6.], [3., -4.]])
    idx = torch.tensor([0, 0, 2, 0])
    s = sage_ref.scatter_ref(x_j, idx, 4, "sum")
    assert s.tolist() == [[7., -10.], [0., 0.], [-5., -6.], [0., 0.]]
    m = sage_ref.scatter_ref(x_j, idx, 4, "mean")
    assert torch.allclose(m, torch.tensor([[7 / 3, -10 / 3], [0., 0.], [-5., -6.], [0., 0.]]))
    mx = sage_ref.scatter_ref(x_j, idx, 4, "max")
    # all-negative column stays negative; untouched rows are 0
    assert mx.tolist() == [[3., -2.], [0., 0.], [-5., -6.], [0., 0.]]


def test_sage_forward_hand_example():
    x = torch.tensor([[1., 0.], [0., 2.], [2., 2.]])
    ei = torch.tensor([[1, 2, 0], [0, 0, 1]])        # 0 <- {1,2}, 1 <- {0}, 2 <- {}
    wl = torch.tensor([[1., 0.], [0., 1.]])
    wr = torch.tensor([[0., 1.], [1., 0.]])
    out = sage_ref.sage_forward(x, ei, wl, wr, "mean", normalize=False)
    # agg = [[1,2],[1,0],[0,0]]; x @ wr^T = [[0,1],[2,0],[2,2]]
    assert out.tolist() == [[1., 3.], [3., 0.], [2., 2.]]
    outn = sage_ref.sage_forward(x, ei, wl, wr, "mean", normalize=True)
    assert torch.allclose(outn, out / out.norm(dim=1, keepdim=True))
    # gcn: mean over neighbours + self for targets with an in-edge, no root term
    og = sage_ref.sage_forward(x, ei, wl, None, "gcn", normalize=False)
    assert torch.allclose(og, torch.tensor([[1., 4 / 3], [0.5, 1.], [0., 0.]]))
    # max_pool with identity pool weight and zero bias = max over relu(x_j)
    om = sage_ref.sage_forward(x, ei, wl, wr, "max_pool", pool_w=torch.eye(2), pool_b=torch.zeros(2),
                               normalize=False)
    assert om.tolist() == [[2., 3.], [3., 0.], [2., 2.]]


def test_adam_ref_matches_torch():
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(50, generator=g)
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=1e-2, weight_decay=0.01)
    pr, m, v = p0.clone(), torch.zeros(50), torch.zeros(50)
    for step in range(1, 6):
        grad = torch.randn(50, generator=g)
        p.grad = grad.clone()
        opt.step()
        pr, m, v = sage_ref.adam_ref(pr, grad, m, v, step, 1e-2, weight_decay=0.01)
        assert torch.allclose(p.detach(), pr, rtol=1e-6, atol=1e-7)


def test_pruned_equals_unpruned_on_seed_rows():
    """The rows a 2-layer model keeps ([:B]) depend only on hop-1 targets' layer-1 rows."""
    from tests.helpers import make_layers, random_graph
    ei = random_graph(300, 3000, seed=1, isolated=5)
    colptr, row, _ = csc.to_csc(ei, 300)
    seeds = np.arange(0, 40)
    node, src, dst, eid, hop_nodes, hop_edges = sampler.neighbor_sample(colptr, row, seeds, [5, 3], 1, 2)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(300, 12, generator=g)
    layers = make_layers([12, 8, 4], "mean", 3)
    xb, _, eib = sage_ref.batch_assemble(x, None, node, src, dst)
    full = sage_ref.graphsage_forward(xb, eib, layers, "mean")[:40]
    # pruned: layer 1 on the first hop_nodes[1] rows with all edges, layer 2 on seeds with hop-1 edges
    t1, e1 = hop_nodes[1], hop_edges[1]
    h1 = sage_ref.sage_forward(xb, eib, layers[0]["lin_l"], layers[0]["lin_r"], "mean")[:t1].relu()
    ei2 = eib[:, :e1]
    out = sage_ref.sage_forward(h1, ei2, layers[1]["lin_l"], layers[1]["lin_r"], "mean")[:40]
    assert torch.allclose(full, out, atol=1e-6)


def test_lstm_aggregate_ref_uses_torch_lstmcell_semantics():
    """The oracle's zero-state cell equals torch.nn.LSTMCell on the flattened, padded input."""
    g = torch.Generator().manual_seed(0)
    Fdim, n_in = 4, 3
    cell = torch.nn.LSTMCell(n_in * Fdim, Fdim)
    x = torch.randn(5, Fdim, generator=g)
    ei = torch.tensor([[1, 2, 3, 4, 0], [0, 0, 0, 0, 2]])      # 0 <- {1,2,3,4} (truncated to 3), 2 <- {0} (padded)
    out = sage_ref.lstm_aggregate_ref(x, ei, 5, n_in, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)
    h0, _ = cell(torch.cat([x[1], x[2], x[3]]).unsqueeze(0))
    h2, _ = cell(torch.cat([x[0], x[0], x[0]]).unsqueeze(0))
    assert torch.allclose(out[0], h0.squeeze(0), atol=1e-6) and torch.allclose(out[2], h2.squeeze(0), atol=1e-6)
    assert bool((out[1] == 0).all()) and bool((out[3] == 0).all())


def test_random_nodes_range_and_uniformity():
    a = sampler.random_nodes(1000, 5000, 3, 7)
    assert a.min() >= 0 and a.max() < 1000 and a.dtype == np.int64
    assert a.tolist() == sampler.random_nodes(1000, 5000, 3, 7).tolist()          # a pure function
    assert a[:100].tolist() == sampler.random_nodes(1000, 100, 3, 7).tolist()     # prefix-consistent
    assert a.tolist() != sampler.random_nodes(1000, 5000, 3, 8).tolist()
    counts = np.bincount(a // 100, minlength=10)                                   # 10 bins, expect 500 each
    assert ((counts - 500.0) ** 2 / 500.0).sum() < 30.0                            # chi^2, 9 dof
