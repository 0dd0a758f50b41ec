"""HIP path against the committed golden vectors (tests/golden/*.npz): CSC build and sampler
bit-exact; forward / loss / gradients / one Adam step of the fig2a step within fp32 tolerances
(logits atol 2e-5, gradients rtol 1e-3, parameters after Adam atol 1e-6)."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SAMPLER_FILES = sorted(glob.glob(os.path.join(GOLD, "sampler_*.npz")))
MODEL_FILES = sorted(glob.glob(os.path.join(GOLD, "model_small_*.npz")))


def _sample(g, dev):
    from graphsage_amd import ops
    n = int(g["num_nodes"])
    ei = torch.from_numpy(g["edge_index"].astype(np.int64)).to(dev)
    colptr, row, perm = ops.csc_build(ei, n)
    seeds = torch.from_numpy(g["seeds"]).to(dev)
    fan = g["fanout"].tolist()
    B = seeds.numel()
    node_cap, edge_cap, frontier = B, 0, B
    for k in fan:
        edge_cap += frontier * k
        frontier = min(frontier * k, n)
        node_cap += frontier
    buf = ops.SampleBuffers(n, node_cap, edge_cap, dev)
    table = ops.new_relabel_table(n, dev)
    ops.sample_fanout(colptr, row, seeds, fan, int(g["rng_seed"]), int(g["counter"]), table, 1, buf)
    return colptr, row, perm, buf


@pytest.mark.parametrize("path", SAMPLER_FILES, ids=os.path.basename)
def test_csc_and_sampler_match_golden(dev, path):
    from graphsage_amd import ops
    g = np.load(path)
    colptr, row, perm, buf = _sample(g, dev)
    assert colptr.cpu().numpy().tolist() == g["colptr"].tolist()
    assert row.cpu().numpy().tolist() == g["row"].tolist()
    assert perm.cpu().numpy().tolist() == g["perm"].tolist()
    st = buf.state.tolist()
    nn, ne = st[ops.ST_NODES], st[ops.ST_EDGES]
    assert st[ops.ST_ERROR] == 0
    assert buf.node[:nn].cpu().numpy().tolist() == g["node"].tolist()
    assert buf.src[:ne].cpu().numpy().tolist() == g["src"].tolist()
    assert buf.dst[:ne].cpu().numpy().tolist() == g["dst"].tolist()
    assert buf.eid[:ne].cpu().numpy().tolist() == g["eid"].tolist()
    hops = len(g["fanout"])
    assert st[ops.ST_HOP_NODES:ops.ST_HOP_NODES + hops + 1] == g["hop_nodes"].tolist()


@pytest.mark.parametrize("path", MODEL_FILES, ids=os.path.basename)
def test_step_matches_golden(dev, path):
    from graphsage_amd.data import Data
    from graphsage_amd.models import GraphSAGE
    agg = os.path.basename(path)[len("model_small_"):-len(".npz")]
    g = np.load(path)
    s = np.load(os.path.join(GOLD, "sampler_small.npz"))
    dims = g["dims"].tolist()
    data = Data(x=torch.from_numpy(g["x"]), y=torch.from_numpy(g["y"]),
                edge_index=torch.from_numpy(s["edge_index"].astype(np.int64)))
    seeds = s["seeds"]
    model = GraphSAGE(dims[0], dims[1], agg, len(dims) - 1, dims[-1]).to(dev)
    with torch.no_grad():
        for i, m in enumerate(model.layers):
            m.lin_l.weight.copy_(torch.from_numpy(g[f"w{i}_lin_l"]))
            if f"w{i}_lin_r" in g.files:
                m.lin_r.weight.copy_(torch.from_numpy(g[f"w{i}_lin_r"]))
            if f"w{i}_pool_w" in g.files:
                m.pool.weight.copy_(torch.from_numpy(g[f"w{i}_pool_w"]))
                m.pool.bias.copy_(torch.from_numpy(g[f"w{i}_pool_b"]))
    # the batch is the GOLDEN one: the stored local edge list, as a plain [2,E] tensor
    ei = torch.from_numpy(np.stack([s["src"], s["dst"]]).astype(np.int64)).to(dev)
    node = torch.from_numpy(s["node"].astype(np.int64)).to(dev)
    x = data.x.to(dev)[node]
    y = data.y.to(dev)[node]
    B = len(seeds)
    opt = torch.optim.Adam(model.parameters(), lr=float(g["lr"]))
    y_hat = model(x, ei)[:B]
    loss = F.cross_entropy(y_hat, y[:B])
    loss.backward()
    np.testing.assert_allclose(y_hat.detach().cpu().numpy(), g["y_hat"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-5)
    names = {"lin_l": lambda m: m.lin_l.weight, "lin_r": lambda m: m.lin_r.weight,
             "pool_w": lambda m: m.pool.weight, "pool_b": lambda m: m.pool.bias}
    for i, m in enumerate(model.layers):
        for k, get in names.items():
            if f"g{i}_{k}" in g.files:
                np.testing.assert_allclose(get(m).grad.cpu().numpy(), g[f"g{i}_{k}"], rtol=1e-3, atol=1e-6)
    opt.step()
    for i, m in enumerate(model.layers):
        for k, get in names.items():
            if f"a{i}_{k}" in g.files:
                np.testing.assert_allclose(get(m).detach().cpu().numpy(), g[f"a{i}_{k}"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("agg", ["mean", "max", "gcn", "max_pool", "max_pool/pull"])
def test_engine_step_matches_golden(dev, agg, monkeypatch):
    """The static-shape engine (sampler + fused forward/backward + device Adam) on the golden batch.
    max_pool: layer 0's edge pass pushed from the targets (what a batch this small takes) and,
    "/pull", pulled by the source rows from the per-source edge lists (what Reddit-sized batches take)."""
    agg, _, pull = agg.partition("/")
    if pull:
        monkeypatch.setenv("SAGE_POOL_PULL_ALL", "1")
    from graphsage_amd.data import Data
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import NeighborSampler
    from graphsage_amd.samplers.uniform_loader import device_graph
    g = np.load(os.path.join(GOLD, f"model_small_{agg}.npz"))
    s = np.load(os.path.join(GOLD, "sampler_small.npz"))
    dims = g["dims"].tolist()
    data = Data(x=torch.from_numpy(g["x"]), y=torch.from_numpy(g["y"]),
                edge_index=torch.from_numpy(s["edge_index"].astype(np.int64)))
    model = GraphSAGE(dims[0], dims[1], agg, 2, dims[-1]).to(dev)
    with torch.no_grad():
        for i, m in enumerate(model.layers):
            m.lin_l.weight.copy_(torch.from_numpy(g[f"w{i}_lin_l"]))
            if agg != "gcn":
                m.lin_r.weight.copy_(torch.from_numpy(g[f"w{i}_lin_r"]))
            if agg == "max_pool":
                m.pool.weight.copy_(torch.from_numpy(g[f"w{i}_pool_w"]))
                m.pool.bias.copy_(torch.from_numpy(g[f"w{i}_pool_b"]))
    graph = device_graph(data)
    B = len(s["seeds"])
    engine = SageTrainEngine(model, NeighborSampler(data, s["fanout"].tolist(), seed=int(s["rng_seed"])),
                             graph.features(data), graph.labels(data), B, lr=float(g["lr"]), keep_grads=True)
    assert (engine.rev is not None) == bool(pull)
    loss = engine.step(torch.from_numpy(s["seeds"]).to(dev), counter=int(s["counter"]))
    nn = int(engine.buf.state[0])
    assert engine.buf.node[:nn].cpu().numpy().tolist() == s["node"].tolist()
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-5)
    np.testing.assert_allclose(engine.act[-1]["y"][:B].cpu().numpy(), g["y_hat"], rtol=1e-4, atol=2e-5)
    for i, m in enumerate(model.layers):
        if agg == "max_pool":
            for k, q in (("pool_w", m.pool.weight), ("pool_b", m.pool.bias)):
                np.testing.assert_allclose(engine.gviews[q].cpu().numpy(), g[f"g{i}_{k}"], rtol=1e-3, atol=1e-6)
                np.testing.assert_allclose(q.detach().cpu().numpy(), g[f"a{i}_{k}"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(engine.gviews[m.lin_l.weight].cpu().numpy(), g[f"g{i}_lin_l"], rtol=1e-3, atol=1e-6)
        np.testing.assert_allclose(m.lin_l.weight.detach().cpu().numpy(), g[f"a{i}_lin_l"], rtol=1e-4, atol=1e-6)
        if agg == "gcn":
            assert not hasattr(m, "lin_r")
            continue
        np.testing.assert_allclose(engine.gviews[m.lin_r.weight].cpu().numpy(), g[f"g{i}_lin_r"], rtol=1e-3, atol=1e-6)
        np.testing.assert_allclose(m.lin_r.weight.detach().cpu().numpy(), g[f"a{i}_lin_r"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("unsup,tag", [(True, "both"), ("only", "unsup")])
def test_engine_unsup_step_matches_golden(dev, unsup, tag):
    """The trainers' steps on the engine against unsup_small.npz: positive / negative ids
    bit-exact, loss rtol 1e-5, gradients rtol 1e-3."""
    from graphsage_amd.data import Data
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import NeighborSampler
    from graphsage_amd.samplers.uniform_loader import device_graph
    g = np.load(os.path.join(GOLD, "model_small_mean.npz"))
    s = np.load(os.path.join(GOLD, "sampler_small.npz"))
    u = np.load(os.path.join(GOLD, "unsup_small.npz"))
    dims = g["dims"].tolist()
    data = Data(x=torch.from_numpy(g["x"]), y=torch.from_numpy(g["y"]),
                edge_index=torch.from_numpy(s["edge_index"].astype(np.int64)))
    model = GraphSAGE(dims[0], dims[1], "mean", 2, dims[-1]).to(dev)
    with torch.no_grad():
        for i, m in enumerate(model.layers):
            m.lin_l.weight.copy_(torch.from_numpy(g[f"w{i}_lin_l"]))
            m.lin_r.weight.copy_(torch.from_numpy(g[f"w{i}_lin_r"]))
    graph = device_graph(data)
    B = len(s["seeds"])
    engine = SageTrainEngine(model, NeighborSampler(data, s["fanout"].tolist(), seed=int(s["rng_seed"])),
                             graph.features(data), graph.labels(data), B, lr=float(u["lr"]), keep_grads=True,
                             unsup=unsup)
    loss = engine.step(torch.from_numpy(s["seeds"]).to(dev), counter=int(s["counter"]))
    nn = int(engine.buf.state[0])
    assert engine.pos_id[:nn].cpu().numpy().tolist() == u["pos"].tolist()
    assert engine.neg_id[:nn].cpu().numpy().tolist() == u["neg"].tolist()
    np.testing.assert_allclose(float(loss), float(u[f"loss_{tag}"]), rtol=1e-5)
    for i, m in enumerate(model.layers):
        np.testing.assert_allclose(engine.gviews[m.lin_l.weight].cpu().numpy(), u[f"g_{tag}_{i}_lin_l"],
                                   rtol=1e-3, atol=1e-6)
        np.testing.assert_allclose(engine.gviews[m.lin_r.weight].cpu().numpy(), u[f"g_{tag}_{i}_lin_r"],
                                   rtol=1e-3, atol=1e-6)


def test_engine_multilabel_step_matches_golden(dev):
    """multilabel_small.npz: the engine with fp32 [N, C] targets (sage_logits_bce_fused as the last
    layer's loss; C = 70: more outputs than a wave has lanes) — sampled batch bit-exact, loss
    rtol 1e-5, gradients rtol 1e-3, parameters after Adam, and evaluate()'s (tp, fp, fn)."""
    from graphsage_amd.data import Data
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import NeighborSampler
    from graphsage_amd.samplers.uniform_loader import device_graph
    g = np.load(os.path.join(GOLD, "multilabel_small.npz"))
    s = np.load(os.path.join(GOLD, "sampler_small.npz"))
    dims = g["dims"].tolist()
    data = Data(x=torch.from_numpy(g["x"]), y=torch.from_numpy(g["y"]),
                edge_index=torch.from_numpy(s["edge_index"].astype(np.int64)))
    model = GraphSAGE(dims[0], dims[1], "mean", 2, dims[-1]).to(dev)
    with torch.no_grad():
        for i, m in enumerate(model.layers):
            m.lin_l.weight.copy_(torch.from_numpy(g[f"w{i}_lin_l"]))
            m.lin_r.weight.copy_(torch.from_numpy(g[f"w{i}_lin_r"]))
    graph = device_graph(data)
    B = len(s["seeds"])
    seeds = torch.from_numpy(s["seeds"]).to(dev)
    engine = SageTrainEngine(model, NeighborSampler(data, s["fanout"].tolist(), seed=int(s["rng_seed"])),
                             graph.features(data), graph.labels(data), B, lr=float(g["lr"]), keep_grads=True)
    assert engine.multilabel
    # evaluation first (forward only, weights untouched): counter0 = the golden batch's counter
    tp, fp, fn = engine.evaluate(seeds, counter0=int(s["counter"]))
    assert [int(tp), int(fp), int(fn)] == g["stats"].tolist()
    loss = engine.step(seeds, counter=int(s["counter"]))
    nn = int(engine.buf.state[0])
    assert engine.buf.node[:nn].cpu().numpy().tolist() == s["node"].tolist()
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-5)
    np.testing.assert_allclose(engine.act[-1]['y'][:B].cpu().numpy(), g["y_hat"], rtol=1e-4, atol=2e-5)
    for i, m in enumerate(model.layers):
        for k, p in (("lin_l", m.lin_l.weight), ("lin_r", m.lin_r.weight)):
            np.testing.assert_allclose(engine.gviews[p].cpu().numpy(), g[f"g{i}_{k}"], rtol=1e-3, atol=1e-6)
            np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"a{i}_{k}"], rtol=1e-4, atol=1e-5)
