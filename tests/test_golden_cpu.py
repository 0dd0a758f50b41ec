"""The committed golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py) pin the
CPU oracle: the numpy sampler, the C sampler, to_csc and the torch restatement of the fig2a step
must keep reproducing them.  Parity with the reference itself stays unpinned."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import csc, sage_ref, sampler, sampler_c

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SAMPLER_FILES = sorted(glob.glob(os.path.join(GOLD, "sampler_*.npz")))
MODEL_FILES = sorted(glob.glob(os.path.join(GOLD, "model_small_*.npz")))


def test_golden_files_present():
    assert len(SAMPLER_FILES) == 3 and len(MODEL_FILES) == 4


@pytest.mark.parametrize("path", SAMPLER_FILES, ids=os.path.basename)
def test_oracle_sampler_reproduces_golden(path):
    g = np.load(path)
    n = int(g["num_nodes"])
    colptr, row, perm = csc.to_csc(g["edge_index"].astype(np.int64), n)
    assert colptr.tolist() == g["colptr"].tolist()
    assert row.tolist() == g["row"].tolist() and perm.tolist() == g["perm"].tolist()
    fan = g["fanout"].tolist()
    impls = [sampler.neighbor_sample]
    if sampler_c.available():
        impls.append(sampler_c.neighbor_sample)
    for fn in impls:
        node, src, dst, eid, hop_nodes, hop_edges = fn(colptr, row, g["seeds"], fan, int(g["rng_seed"]),
                                                       int(g["counter"]))
        assert node.tolist() == g["node"].tolist()
        assert src.tolist() == g["src"].tolist() and dst.tolist() == g["dst"].tolist()
        assert eid.tolist() == g["eid"].tolist()
        assert list(hop_nodes) == g["hop_nodes"].tolist() and list(hop_edges) == g["hop_edges"].tolist()
    # structural invariants of the stored vectors themselves (SURVEY §8c sampler KATs)
    node, src, dst, eid = (g[k].astype(np.int64) for k in ("node", "src", "dst", "eid"))
    B = len(g["seeds"])
    assert node[:B].tolist() == g["seeds"].tolist()
    assert len(set(node[B:].tolist())) == len(node) - B
    assert (row[eid] == node[src]).all()
    owner = np.searchsorted(colptr, eid, side="right") - 1
    assert (owner == node[dst]).all()
    assert (np.diff(dst) >= 0).all()                       # grouped by target, frontier order


@pytest.mark.parametrize("path", MODEL_FILES, ids=os.path.basename)
def test_oracle_step_reproduces_golden(path):
    agg = os.path.basename(path)[len("model_small_"):-len(".npz")]
    g = np.load(path)
    s = np.load(os.path.join(GOLD, "sampler_small.npz"))
    n_layers = len(g["dims"]) - 1
    layers = []
    for i in range(n_layers):
        layers.append({k[len(f"w{i}_"):]: torch.from_numpy(g[k]).clone().requires_grad_(True)
                       for k in g.files if k.startswith(f"w{i}_")})
    t = lambda k: torch.from_numpy(s[k].astype(np.int64))
    loss, y_hat = sage_ref.train_step_loss(torch.from_numpy(g["x"]), torch.from_numpy(g["y"]), t("node"), t("src"),
                                           t("dst"), len(s["seeds"]), layers, agg)
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-6)
    np.testing.assert_allclose(y_hat.detach().numpy(), g["y_hat"], rtol=1e-5, atol=1e-6)
    for i, p in enumerate(layers):
        for k, v in p.items():
            np.testing.assert_allclose(v.grad.numpy(), g[f"g{i}_{k}"], rtol=1e-4, atol=1e-7)


def test_oracle_unsup_step_reproduces_golden():
    """unsup_small.npz: positive / negative ids bit-exact, the unsupervised loss (alone and added to
    the cross entropy) and its gradients, on the batch of sampler_small.npz with the weights of
    model_small_mean.npz."""
    u = np.load(os.path.join(GOLD, "unsup_small.npz"))
    s = np.load(os.path.join(GOLD, "sampler_small.npz"))
    m = np.load(os.path.join(GOLD, "model_small_mean.npz"))
    n = int(s["num_nodes"])
    colptr, row, _ = csc.to_csc(s["edge_index"].astype(np.int64), n)
    node = s["node"].astype(np.int64)
    seed, counter = int(s["rng_seed"]), int(s["counter"])
    pos = sampler.random_neighbor(colptr, row, node, seed, counter)
    neg = sampler.random_nodes(n, len(node), seed, counter)
    assert pos.tolist() == u["pos"].tolist() and neg.tolist() == u["neg"].tolist()
    has = colptr[node + 1] > colptr[node]
    assert all(p in row[colptr[v]:colptr[v + 1]] for v, p in zip(node[has], pos[has]))   # an in-neighbour
    assert (pos[~has] == node[~has]).all()                                            # isolated: itself
    x, y = torch.from_numpy(m["x"]), torch.from_numpy(m["y"])
    B = len(s["seeds"])
    ei = torch.from_numpy(np.stack([s["src"].astype(np.int64), s["dst"].astype(np.int64)]))
    for tag in ("unsup", "both"):
        layers = [{k[len(f"w{i}_"):]: torch.from_numpy(m[k]).clone().requires_grad_(True)
                   for k in m.files if k.startswith(f"w{i}_")} for i in range(2)]
        outs = [sage_ref.graphsage_forward(x[torch.from_numpy(ids)], ei, layers)[:B] for ids in (node, pos, neg)]
        loss = sage_ref.unsup_loss(*outs)
        if tag == "both":
            loss = loss + torch.nn.functional.cross_entropy(outs[0], y[torch.from_numpy(node[:B])])
        loss.backward()
        np.testing.assert_allclose(float(loss.detach()), float(u[f"loss_{tag}"]), rtol=1e-6)
        for i, p in enumerate(layers):
            for k, v in p.items():
                np.testing.assert_allclose(v.grad.numpy(), u[f"g_{tag}_{i}_{k}"], rtol=1e-4, atol=1e-7)


def test_oracle_multilabel_step_reproduces_golden():
    """multilabel_small.npz (BASELINE configs[1] in small): BCE-with-logits on the normalised
    logits of the seed rows, its gradients, and the (tp, fp, fn) of the prediction y > 0."""
    g = np.load(os.path.join(GOLD, "multilabel_small.npz"))
    s = np.load(os.path.join(GOLD, "sampler_small.npz"))
    layers = [{k[len(f"w{i}_"):]: torch.from_numpy(g[k]).clone().requires_grad_(True)
               for k in g.files if k.startswith(f"w{i}_")} for i in range(2)]
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    node = torch.from_numpy(s["node"].astype(np.int64))
    ei = torch.from_numpy(np.stack([s["src"].astype(np.int64), s["dst"].astype(np.int64)]))
    B = len(s["seeds"])
    y_hat = sage_ref.graphsage_forward(x[node], ei, layers, "mean")[:B]
    loss = torch.nn.functional.binary_cross_entropy_with_logits(y_hat, y[node[:B]])
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-6)
    np.testing.assert_allclose(y_hat.detach().numpy(), g["y_hat"], rtol=1e-5, atol=1e-6)
    for i, p in enumerate(layers):
        for k, v in p.items():
            np.testing.assert_allclose(v.grad.numpy(), g[f"g{i}_{k}"], rtol=1e-4, atol=1e-7)
    p_, t_ = y_hat.detach() > 0, y[node[:B]] > 0.5
    assert [int((p_ & t_).sum()), int((p_ & ~t_).sum()), int((~p_ & t_).sum())] == g["stats"].tolist()
    assert set(np.unique(g["y"]).tolist()) == {0.0, 1.0} and g["y"].shape[1] == 70      # more outputs than one wave has lanes
