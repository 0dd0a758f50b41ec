#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ from the CPU oracle.

The reference cannot run here (its torch_geometric / torch_sparse / torch_scatter dependencies
are absent) and holds no fixtures of its own, so these vectors pin the ORACLE, not
the reference: they freeze today's oracle outputs so that (a) a later edit of oracle/ that changes
a result is caught by tests/test_golden_cpu.py and (b) the HIP path is checked against files,
not only against an oracle evaluated in the same process (tests/test_gpu_golden.py).

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz (deterministic)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import csc, sage_ref, sampler  # noqa: E402
from tests.helpers import make_layers, random_graph  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def sampler_case(n, e, iso, hubs, B, fanout, graph_seed, rng_seed, counter, dup=False):
    ei = random_graph(n, e, graph_seed, isolated=iso, hubs=hubs)
    colptr, row, perm = csc.to_csc(ei, n)
    rng = np.random.default_rng(graph_seed + 100)
    seeds = rng.choice(n, size=B, replace=False)
    if iso:
        seeds[0] = 0
    if dup:
        seeds[-1] = seeds[1]
    node, src, dst, eid, hop_nodes, hop_edges = sampler.neighbor_sample(colptr, row, seeds, fanout, rng_seed, counter)
    return dict(edge_index=ei.astype(np.int32), num_nodes=np.int64(n), colptr=colptr, row=row.astype(np.int32),
                perm=perm.astype(np.int32), seeds=seeds.astype(np.int64), fanout=np.array(fanout, dtype=np.int64),
                rng_seed=np.int64(rng_seed), counter=np.int64(counter), node=node.astype(np.int32),
                src=src.astype(np.int32), dst=dst.astype(np.int32), eid=eid.astype(np.int32),
                hop_nodes=np.array(hop_nodes, dtype=np.int64), hop_edges=np.array(hop_edges, dtype=np.int64))


def model_case(case, aggregator, dims, feat_seed, layer_seed, lr=1e-2):
    n = int(case["num_nodes"])
    g = torch.Generator().manual_seed(feat_seed)
    x = torch.randn(n, dims[0], generator=g)
    y = torch.randint(0, dims[-1], (n,), generator=g)
    layers = make_layers(dims, aggregator, layer_seed)
    leaf = [{k: v.clone().requires_grad_(True) for k, v in p.items()} for p in layers]
    B = len(case["seeds"])
    node = torch.from_numpy(case["node"].astype(np.int64))
    src = torch.from_numpy(case["src"].astype(np.int64))
    dst = torch.from_numpy(case["dst"].astype(np.int64))
    loss, y_hat = sage_ref.train_step_loss(x, y, node, src, dst, B, leaf, aggregator)
    loss.backward()
    out = dict(x=x.numpy(), y=y.numpy(), loss=np.float32(loss.detach()), y_hat=y_hat.detach().numpy(),
               dims=np.array(dims, dtype=np.int64), lr=np.float32(lr))
    for i, (p, lf) in enumerate(zip(layers, leaf)):
        for k in p:
            out[f"w{i}_{k}"] = p[k].numpy()
            out[f"g{i}_{k}"] = lf[k].grad.numpy()
            new, _, _ = sage_ref.adam_ref(p[k], lf[k].grad, torch.zeros_like(p[k]), torch.zeros_like(p[k]), 1, lr)
            out[f"a{i}_{k}"] = new.numpy()      # parameter after one Adam step from zero state
    return out


def unsup_case(case, dims, feat_seed, layer_seed, lr=1e-2):
    """The trainers' step on the golden batch (mean aggregator): positive / negative node ids
    (samplers/utils.py:7-37 restated with the build's Philox counters), the unsupervised loss
    alone and supervised + unsupervised, with their weight gradients."""
    n = int(case["num_nodes"])
    g = torch.Generator().manual_seed(feat_seed)
    x = torch.randn(n, dims[0], generator=g)
    y = torch.randint(0, dims[-1], (n,), generator=g)
    layers = make_layers(dims, "mean", layer_seed)
    ei = case["edge_index"].astype(np.int64)
    colptr, row, _ = csc.to_csc(ei, n)
    node = case["node"].astype(np.int64)
    rng_seed, counter = int(case["rng_seed"]), int(case["counter"])
    pos = sampler.random_neighbor(colptr, row, node, rng_seed, counter)
    neg = sampler.random_nodes(n, len(node), rng_seed, counter)
    B = len(case["seeds"])
    e2 = torch.from_numpy(np.stack([case["src"].astype(np.int64), case["dst"].astype(np.int64)]))
    out = dict(pos=pos.astype(np.int32), neg=neg.astype(np.int32), dims=np.array(dims, dtype=np.int64),
               lr=np.float32(lr))
    for tag, with_sup in (("unsup", False), ("both", True)):
        leaf = [{k: v.clone().requires_grad_(True) for k, v in p.items()} for p in layers]
        outs = [sage_ref.graphsage_forward(x[torch.from_numpy(ids)], e2, leaf)[:B] for ids in (node, pos, neg)]
        loss = sage_ref.unsup_loss(*outs)
        if with_sup:
            loss = loss + torch.nn.functional.cross_entropy(outs[0], y[torch.from_numpy(node[:B])])
        loss.backward()
        out[f"loss_{tag}"] = np.float32(loss.detach())
        for i, lf in enumerate(leaf):
            for k in lf:
                out[f"g_{tag}_{i}_{k}"] = lf[k].grad.numpy()
    return out


def multilabel_case(case, dims, feat_seed, layer_seed, lr=1e-2):
    """BASELINE configs[1] in small: the golden batch with multi-label targets (fp32 0 / 1, C
    independent sigmoid outputs) and BCEWithLogitsLoss on the normalised logits of the seed rows
    (reference examples/ppi_graphsage_supervised.py:30): loss, gradients, parameters after one
    Adam step, and micro-F1's ingredients for the prediction y > 0
    (graph_level_graphsage_trainers.py:57-58)."""
    n = int(case["num_nodes"])
    g = torch.Generator().manual_seed(feat_seed)
    x = torch.randn(n, dims[0], generator=g)
    y = (torch.rand(n, dims[-1], generator=g) < 0.4).float()
    layers = make_layers(dims, "mean", layer_seed)
    leaf = [{k: v.clone().requires_grad_(True) for k, v in p.items()} for p in layers]
    B = len(case["seeds"])
    node = torch.from_numpy(case["node"].astype(np.int64))
    ei = torch.from_numpy(np.stack([case["src"].astype(np.int64), case["dst"].astype(np.int64)]))
    y_hat = sage_ref.graphsage_forward(x[node], ei, leaf, "mean")[:B]
    loss = torch.nn.functional.binary_cross_entropy_with_logits(y_hat, y[node[:B]])
    loss.backward()
    p_, t_ = y_hat.detach() > 0, y[node[:B]] > 0.5
    out = dict(x=x.numpy(), y=y.numpy(), loss=np.float32(loss.detach()), y_hat=y_hat.detach().numpy(),
               dims=np.array(dims, dtype=np.int64), lr=np.float32(lr),
               stats=np.array([int((p_ & t_).sum()), int((p_ & ~t_).sum()), int((~p_ & t_).sum())], dtype=np.int64))
    for i, (p, lf) in enumerate(zip(layers, leaf)):
        for k in p:
            out[f"w{i}_{k}"] = p[k].numpy()
            out[f"g{i}_{k}"] = lf[k].grad.numpy()
            new, _, _ = sage_ref.adam_ref(p[k], lf[k].grad, torch.zeros_like(p[k]), torch.zeros_like(p[k]), 1, lr)
            out[f"a{i}_{k}"] = new.numpy()
    return out


def main():
    small = sampler_case(300, 2400, 5, 2, 40, [4, 3], graph_seed=5, rng_seed=21, counter=9, dup=True)
    np.savez_compressed(os.path.join(HERE, "sampler_small.npz"), **small)
    # Cora-SHAPED synthetic graph (N=2,708, E=10,556 directed, 140 seeds, fan-out 25/10: mostly take-all)
    cora = sampler_case(2708, 10556, 0, 0, 140, [25, 10], graph_seed=17, rng_seed=0, counter=0)
    np.savez_compressed(os.path.join(HERE, "sampler_cora_shape.npz"), **cora)
    # a dense little graph where fan-out < degree almost everywhere (Floyd path) and one 70-wide hop
    dense = sampler_case(400, 40000, 3, 1, 24, [70, 5], graph_seed=23, rng_seed=3, counter=123456789012)
    np.savez_compressed(os.path.join(HERE, "sampler_dense.npz"), **dense)
    for agg in ("mean", "max", "max_pool", "gcn"):
        m = model_case(small, agg, [12, 8, 4], feat_seed=31, layer_seed=32)
        np.savez_compressed(os.path.join(HERE, f"model_small_{agg}.npz"), **m)
    # same features / labels / weights as model_small_mean.npz (feat_seed 31, layer_seed 32)
    np.savez_compressed(os.path.join(HERE, "unsup_small.npz"), **unsup_case(small, [12, 8, 4], 31, 32))
    np.savez_compressed(os.path.join(HERE, "multilabel_small.npz"), **multilabel_case(small, [12, 8, 70], 41, 42))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
