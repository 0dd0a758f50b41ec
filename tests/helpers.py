"""Shared test inputs (seeded, small enough for the CPU oracle to finish in seconds)."""
import numpy as np
import torch


def random_graph(num_nodes, num_edges, seed, isolated=0, hubs=0):
    """Directed edge list [2,E] with optional nodes of in-degree 0 and a few hubs."""
    rng = np.random.default_rng(seed)
    src = rng.integers(0, num_nodes, num_edges)
    dst = rng.integers(isolated, num_nodes, num_edges)  # nodes < isolated get no in-edges
    if hubs:
        hub_ids = rng.integers(isolated, num_nodes, hubs)
        extra = num_edges // 4
        src = np.concatenate([src, rng.integers(0, num_nodes, extra)])
        dst = np.concatenate([dst, hub_ids[rng.integers(0, hubs, extra)]])
    return np.stack([src, dst]).astype(np.int64)


def make_layers(dims, aggregator, seed):
    """Random layer parameters in the oracle's dict form; dims = [in, hidden, ..., out]."""
    g = torch.Generator().manual_seed(seed)
    layers = []
    for fin, fout in zip(dims[:-1], dims[1:]):
        p = {"lin_l": torch.randn(fout, fin, generator=g) / fin ** 0.5}
        if aggregator != "gcn":
            p["lin_r"] = torch.randn(fout, fin, generator=g) / fin ** 0.5
        if aggregator in ("max_pool", "mean_pool"):
            p["pool_w"] = torch.randn(fin, fin, generator=g) / fin ** 0.5
            p["pool_b"] = torch.randn(fin, generator=g) * 0.1
        layers.append(p)
    return layers


def load_layers(model, layers):
    """Copy oracle-form parameters into a graphsage_amd GraphSAGE / list of SAGE layers."""
    mods = model.layers if hasattr(model, "layers") else model
    with torch.no_grad():
        for m, p in zip(mods, layers):
            m.lin_l.weight.copy_(p["lin_l"])
            if "lin_r" in p:
                m.lin_r.weight.copy_(p["lin_r"])
            if "pool_w" in p:
                m.pool.weight.copy_(p["pool_w"])
                m.pool.bias.copy_(p["pool_b"])


def assert_trajectories_close(got, want, frac=0.02, worst=5e-4):
    """Two engines that ran the same training steps: equal weights up to summation order.  fp32
    atomics arrive in any order, and Adam turns a last-bit difference of a gradient near its eps
    (1e-8: a parameter the batches barely touch) into a visible one — up to lr * |dg| / eps — which
    then rides along for the rest of the trajectory.  So: all but a small fraction of the elements
    agree to rtol 1e-4 / atol 1e-6, and none is further off than `worst` (a lost or doubled update
    moves a weight by ~lr = 1e-2)."""
    diff = (got - want).abs()
    off = diff > 1e-6 + 1e-4 * want.abs()
    assert int(off.sum()) <= frac * off.numel(), (int(off.sum()), off.numel())
    assert float(diff.max()) <= worst, float(diff.max())
