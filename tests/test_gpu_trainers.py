"""f-1 / f-4: trainer loop, evaluation and the fig2a harness on the device (SURVEY §8c end-to-end
KATs: same init + same sampled batches => same loss curve as the CPU oracle for 20 steps;
micro-F1 on a planted-partition graph within 0.5 pt of the oracle-trained model)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import csc, sage_ref, sampler
from tests.helpers import assert_trajectories_close, load_layers, make_layers

pytestmark = pytest.mark.gpu


def _dataset(n=1500, e=30000, f=24, c=5, seed=3, noise=1.5):
    from graphsage_amd.datasets import planted_partition
    from graphsage_amd.datasets.synthetic import SyntheticDataset, _masks
    d = planted_partition(n, e, f, c, homophily=0.85, noise=noise, seed=seed)
    _masks(d, 640, 200, 400, seed)
    return SyntheticDataset([d], c)


def _oracle_train(data, layers, seeds_batches, fanout, rng_seed, counters, lr, B):
    """fig2a step on the CPU oracle: returns the loss curve and the trained layers."""
    c0, r0, _ = csc.to_csc(data.edge_index.numpy(), data.num_nodes)
    params = [{k: v.clone() for k, v in p.items()} for p in layers]
    state = [{k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in p.items()} for p in params]
    losses = []
    for step, (seeds, counter) in enumerate(zip(seeds_batches, counters), start=1):
        node, src, dst, _, _, _ = sampler.neighbor_sample(c0, r0, seeds.numpy(), fanout, rng_seed, counter)
        leaf = [{k: v.clone().requires_grad_(True) for k, v in p.items()} for p in params]
        loss, _ = sage_ref.train_step_loss(data.x, data.y, torch.from_numpy(node), torch.from_numpy(src),
                                           torch.from_numpy(dst), B, leaf)
        loss.backward()
        losses.append(float(loss.detach()))
        for p, lf, st in zip(params, leaf, state):
            for k in p:
                p[k], m, v = sage_ref.adam_ref(p[k], lf[k].grad, st[k][0], st[k][1], step, lr)
                st[k] = (m, v)
    return losses, params


def _oracle_accuracy(data, params, seeds, fanout, rng_seed, counter0, B):
    c0, r0, _ = csc.to_csc(data.edge_index.numpy(), data.num_nodes)
    hit = 0
    for b in range(0, len(seeds), B):
        chunk = seeds[b:b + B]
        node, src, dst, _, _, _ = sampler.neighbor_sample(c0, r0, chunk.numpy(), fanout, rng_seed, counter0 + b // B)
        out = sage_ref.graphsage_forward(data.x[node], torch.from_numpy(np.stack([src, dst])), params)
        hit += int((out[:len(chunk)].argmax(-1) == data.y[chunk]).sum())
    return hit / len(seeds)


def test_engine_loss_curve_and_f1_match_oracle(dev):
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import NeighborSampler
    from graphsage_amd.samplers.uniform_loader import device_graph
    ds = _dataset()
    data = ds[0]
    B, fan, lr = 64, [6, 4], 1e-2
    layers = make_layers([24, 16, 5], "mean", 5)
    model = GraphSAGE(24, 16, "mean", 2, 5).to(dev)
    load_layers(model, layers)
    graph = device_graph(data)
    engine = SageTrainEngine(model, NeighborSampler(data, fan, seed=13), graph.features(data),
                             graph.labels(data), B, lr=lr)
    train = data.train_mask.nonzero().view(-1)
    steps = 20
    batches = [train[(s % 10) * B:(s % 10 + 1) * B] for s in range(steps)]
    counters = list(range(100, 100 + steps))
    want, params = _oracle_train(data, layers, batches, fan, 13, counters, lr, B)
    got = [float(engine.step(b.to(dev), counter=c)) for b, c in zip(batches, counters)]
    # loss curve within 1e-4 relative for 20 steps
    np.testing.assert_allclose(got, want, rtol=1e-4)
    assert want[-1] < want[0]
    # evaluation: same sampled neighbourhoods, same weights to 1e-4 => same predictions up to ties
    test_ids = data.test_mask.nonzero().view(-1)
    correct, count = engine.evaluate(test_ids.to(dev), counter0=7000)
    acc = int(correct) / int(count)
    acc_ref = _oracle_accuracy(data, params, test_ids, fan, 13, 7000, B)
    assert int(count) == 400
    assert abs(acc - acc_ref) <= 0.005, (acc, acc_ref)      # within 0.5 pt micro-F1
    assert acc > 0.5


def test_train_epoch_matches_step_loop(dev):
    from graphsage_amd.dist import batch_counter
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import NeighborSampler
    from graphsage_amd.samplers.uniform_loader import device_graph
    data = _dataset()[0]
    B, fan = 64, [5, 3]
    torch.manual_seed(0)
    model_a = GraphSAGE(24, 16, "sum", 2, 5).to(dev)
    model_b = copy.deepcopy(model_a)
    graph = device_graph(data)
    mk = lambda m: SageTrainEngine(m, NeighborSampler(data, fan, seed=2), graph.features(data),
                                   graph.labels(data), B, lr=1e-2)
    ea, eb = mk(model_a), mk(model_b)
    seeds = data.train_mask.nonzero().view(-1).to(dev)[:5 * B + 17]      # tail of 17 is dropped
    mean_loss = ea.train_epoch(seeds, epoch=3)
    losses = [eb.step(seeds[b * B:(b + 1) * B], counter=batch_counter(3, 5, b, 0, 1)).clone() for b in range(5)]
    torch.testing.assert_close(mean_loss, torch.stack(losses).mean(0), rtol=1e-5, atol=1e-6)
    assert_trajectories_close(ea.flat, eb.flat)
    assert int(ea.step_dev) == 5


@pytest.mark.parametrize("supervised_only,use_engine", [(True, None), (False, None), (False, True)])
def test_supervised_trainer_runs_and_learns(dev, supervised_only, use_engine):
    from graphsage_amd import settings
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import UniformLoader
    from graphsage_amd.trainers import SupervisedGraphSageTrainerForNodeLevelTask, dataloader_kwargs
    ds = _dataset(noise=1.0)
    data = ds[0]
    torch.manual_seed(0)
    model = GraphSAGE(ds.num_features, 32, "mean", 2, ds.num_classes).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    old = dict(dataloader_kwargs)
    dataloader_kwargs['batch_size'] = 128
    try:
        tr = SupervisedGraphSageTrainerForNodeLevelTask(
            data, UniformLoader, loss_fn=F.cross_entropy, model=model, optimizer=opt, num_epochs=3, device=dev,
            k1=6, k2=4, dataset_name='planted', supervised_only=supervised_only, use_engine=use_engine)
    finally:
        dataloader_kwargs.update(old)
    assert (tr.engine is not None) == (supervised_only or bool(use_engine))
    best = tr.run(verbose=False)
    assert set(best) >= {'loss', 'val_acc', 'val_f1', 'test_acc', 'test_f1', 'train_time', 'test_time'}
    assert best['val_f1'] == best['val_acc'] and best['test_f1'] > 0.6
    assert tr.results[-1]['loss'] < tr.results[0]['loss']
    assert len(tr.results) == 3


@pytest.mark.parametrize("use_engine", [False, True])
def test_unsupervised_trainer_runs(dev, use_engine):
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import UniformLoader
    from graphsage_amd.trainers import UnsupervisedGraphSageTrainerForNodeLevelTask, dataloader_kwargs
    ds = _dataset(noise=1.0)
    data = ds[0]
    torch.manual_seed(0)
    model = GraphSAGE(ds.num_features, 32, "mean", 2).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=5e-3)
    old = dict(dataloader_kwargs)
    dataloader_kwargs['batch_size'] = 128
    try:
        tr = UnsupervisedGraphSageTrainerForNodeLevelTask(data, UniformLoader, model=model, optimizer=opt,
                                                          num_epochs=2, device=dev, k1=6, k2=4,
                                                          use_engine=use_engine)
    finally:
        dataloader_kwargs.update(old)
    best = tr.run(verbose=False)
    assert best['test_f1'] > 0.5 and best['loss'] > 0
    assert tr.results[-1]['loss'] < tr.results[0]['loss'] + 0.1


@pytest.mark.parametrize("aggregator", ["mean", "max_pool", "gcn", "lstm"])
def test_fig2a_harness(dev, aggregator):
    from graphsage_amd.experiments import fig2a
    ds = _dataset(noise=1.0)
    if aggregator == "lstm":
        from graphsage_amd import settings
        settings.LSTM_NUM_INPUTS = 4
    out = fig2a.get(aggregator, dataset=ds, num_epochs=2, batch_size=128, k1=6, k2=4, verbose=False).run()
    assert set(out) >= {'test_time', 'train_time'}
    assert out['engine']                                 # every aggregator of the sweep trains on the engine (lstm since round 3)
    assert out['train_time'] > 0 and out['test_time'] > 0 and out['test_acc'] > 0.5


def test_fig2b_harness(dev):
    """Sample-size sweep (reference experiments/fig2b): reference keys, a trained model, and a
    larger neighbourhood sample does not make the test pass cheaper."""
    from graphsage_amd.experiments import fig2b
    ds = _dataset(noise=1.0)
    outs = {s: fig2b.get(s, dataset=ds, num_epochs=4, num_runs=1, hidden_channels=32, learning_rate=1e-2).run() for s in (2, 8)}
    for out in outs.values():
        assert set(out) == {'test_time', 'test_f1'}
        assert out['test_time'] > 0 and out['test_f1'] > 0.5


@pytest.mark.parametrize("unsup,aggregator", [(True, "mean"), ("only", "mean"), (True, "gcn")])
def test_engine_trainer_step_with_unsup_term_matches_oracle(dev, unsup, aggregator):
    """f-2 on the engine: loss = CE + unsupervised term (three forwards over one sampled batch,
    trainers/node_level_graphsage_trainers.py:41 + base_trainers.py:170-181) or the unsupervised
    term alone (:104-117; BASELINE.json configs[4]), gradients and the Adam step against the CPU
    oracle.  Positive / negative ids are bit-exact restatements."""
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import NeighborSampler
    from graphsage_amd.samplers.uniform_loader import device_graph
    data = _dataset()[0]
    B, fan, lr, seed = 48, [5, 4], 1e-2, 17
    layers = make_layers([24, 16, 5], aggregator, 9)
    model = GraphSAGE(24, 16, aggregator, 2, 5).to(dev)
    load_layers(model, layers)
    graph = device_graph(data)
    engine = SageTrainEngine(model, NeighborSampler(data, fan, seed=seed), graph.features(data),
                             graph.labels(data), B, lr=lr, unsup=unsup, keep_grads=True)
    c0, r0, _ = csc.to_csc(data.edge_index.numpy(), data.num_nodes)
    train = data.train_mask.nonzero().view(-1)
    params = [{k: v.clone() for k, v in p.items()} for p in layers]
    state = [{k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in p.items()} for p in params]
    for step in range(1, 4):
        seeds, counter = train[(step - 1) * B:step * B], 200 + step
        loss = engine.step(seeds.to(dev), counter=counter)
        node, src, dst, _, _, _ = sampler.neighbor_sample(c0, r0, seeds.numpy(), fan, seed, counter)
        nb = len(node)
        pos = sampler.random_neighbor(c0, r0, node, seed, counter)
        neg = sampler.random_nodes(data.num_nodes, nb, seed, counter)
        assert engine.pos_id[:nb].cpu().numpy().tolist() == pos.tolist()
        assert engine.neg_id[:nb].cpu().numpy().tolist() == neg.tolist()
        leaf = [{k: v.clone().requires_grad_(True) for k, v in p.items()} for p in params]
        ei = torch.from_numpy(np.stack([src, dst]))
        outs = [sage_ref.graphsage_forward(data.x[torch.from_numpy(ids)], ei, leaf, aggregator)[:B]
                for ids in (node, pos, neg)]
        want = sage_ref.unsup_loss(*outs)
        if unsup is True:
            want = want + F.cross_entropy(outs[0], data.y[torch.from_numpy(node[:B])])
        want.backward()
        torch.testing.assert_close(loss.cpu(), want.detach().view(1), rtol=1e-5, atol=1e-6)
        for m, lf in zip(model.layers, leaf):
            for k in lf:      # lin_l, lin_r (gcn: lin_l only)
                torch.testing.assert_close(engine.gviews[getattr(m, k).weight].cpu(), lf[k].grad, rtol=1e-3, atol=1e-6)
        for p, lf, st in zip(params, leaf, state):
            for k in p:
                p[k], m1, v1 = sage_ref.adam_ref(p[k], lf[k].grad, st[k][0], st[k][1], step, lr)
                st[k] = (m1, v1)
        for m, p in zip(model.layers, params):
            for k in p:
                torch.testing.assert_close(getattr(m, k).weight.detach().cpu(), p[k], rtol=1e-4, atol=1e-6)


def test_engine_partial_last_batch_matches_oracle(dev):
    """The partial last batch of an epoch (the reference's loaders keep it): seeds padded to B,
    loss = mean over the valid rows, padded rows without gradient — against the oracle on the
    same padded seed list with the cross entropy over the valid rows."""
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import NeighborSampler
    from graphsage_amd.samplers.uniform_loader import device_graph
    data = _dataset()[0]
    B, fan, lr, valid = 64, [6, 4], 1e-2, 23
    layers = make_layers([24, 16, 5], "mean", 5)
    model = GraphSAGE(24, 16, "mean", 2, 5).to(dev)
    load_layers(model, layers)
    graph = device_graph(data)
    engine = SageTrainEngine(model, NeighborSampler(data, fan, seed=13), graph.features(data),
                             graph.labels(data), B, lr=lr, keep_grads=True)
    train = data.train_mask.nonzero().view(-1)
    engine.step(train[:B].to(dev), counter=1)                 # a full batch first: buffers are dirty
    params = [{k: getattr(m, k).weight.detach().cpu().clone() for k in ("lin_l", "lin_r")} for m in model.layers]
    rest = train[B:B + valid]
    padded = torch.cat([rest, rest[:1].expand(B - valid)])
    loss = engine.step(padded.to(dev), counter=2, valid=valid)
    c0, r0, _ = csc.to_csc(data.edge_index.numpy(), data.num_nodes)
    node, src, dst, _, _, _ = sampler.neighbor_sample(c0, r0, padded.numpy(), fan, 13, 2)
    leaf = [{k: v.clone().requires_grad_(True) for k, v in p.items()} for p in params]
    out = sage_ref.graphsage_forward(data.x[node], torch.from_numpy(np.stack([src, dst])), leaf)[:valid]
    want = F.cross_entropy(out, data.y[rest])
    want.backward()
    torch.testing.assert_close(loss.cpu().view(()), want.detach(), rtol=1e-5, atol=1e-6)
    for m, lf in zip(model.layers, leaf):
        for k in lf:
            torch.testing.assert_close(engine.gviews[getattr(m, k).weight].cpu(), lf[k].grad, rtol=1e-3, atol=1e-6)
    # and the next full batch is unaffected by what the partial one left behind
    l3 = float(engine.step(train[:B].to(dev), counter=3))
    assert l3 == l3 and 0 < l3 < 10


def test_fig2a_engine_epochs_reshuffle_and_cover_the_tail(dev):
    """fig2a trains with shuffle=True (fig2a/runners/graphsage_runners.py:72-73): two epochs see
    different batch compositions, and all n_train seeds are trained each epoch (tail included)."""
    from graphsage_amd.experiments import fig2a
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import UniformLoader
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd import ops
    ds = _dataset(noise=1.0)
    data = ds[0]
    loader = UniformLoader(data, input_nodes=data.train_mask, num_neighbors=[6, 4], shuffle=True, batch_size=128)
    o1, o2 = loader._order().clone(), loader._order().clone()
    assert not torch.equal(o1, o2) and torch.equal(o1.sort().values, o2.sort().values)
    torch.manual_seed(0)
    model = GraphSAGE(ds.num_features, 32, "mean", 2, ds.num_classes).to(dev)
    graph = loader.neighbor_sampler.graph
    engine = SageTrainEngine(model, loader.neighbor_sampler, graph.features(data), graph.labels(data), 128, lr=1e-2)
    n_train = int(data.train_mask.sum())                       # 640 = 5 x 128: add a tail by dropping 7 seeds
    loader._idx_dev = loader._idx_dev[:n_train - 7]
    loader.dataset.idx = loader.dataset.idx[:n_train - 7]
    seeds_before = int(engine.state_sum[ops.ST_HOP_NODES])
    losses = [fig2a.train(e, model, None, loader, engine)[0] for e in (1, 2)]
    assert int(engine.step_dev) == 2 * 5                       # 4 full batches + the padded tail, twice
    assert int(engine.state_sum[ops.ST_HOP_NODES]) - seeds_before == 2 * 5 * 128
    assert losses[1] < losses[0]


def test_fig2a_reported_accuracy_matches_oracle(dev, monkeypatch):
    """f-4: what experiments/fig2a reports (test accuracy after its epochs; reference
    experiments/fig2a/runners/graphsage_runners.py:57-66, :92) against the CPU oracle trained on
    the same shuffled batches (the loader's own generator), same Philox counters, and evaluated on
    the same sampled test neighbourhoods: within 0.5 pt."""
    from graphsage_amd.dist import batch_counter
    from graphsage_amd.experiments import fig2a
    from graphsage_amd import settings
    ds = _dataset(noise=1.5)
    data = ds[0]
    B, fan, epochs = 128, [6, 4], 4
    monkeypatch.setattr(fig2a, "LEARNING_RATE", 1e-2)      # 20 steps must move the model (the harness's 1e-3 is for Reddit-length epochs)
    torch.manual_seed(11)
    out = fig2a.get("mean", dataset=ds, num_epochs=epochs, batch_size=B, k1=fan[0], k2=fan[1], verbose=False).run()
    assert out['engine']
    # the oracle's replay of that run: same initial weights (same torch seed), same batch order
    torch.manual_seed(11)
    from graphsage_amd.models import GraphSAGE
    init = GraphSAGE(ds.num_features, fig2a.HIDDEN_CHANNELS, "mean", 2, ds.num_classes)
    layers = [{"lin_l": m.lin_l.weight.detach().clone(), "lin_r": m.lin_r.weight.detach().clone()} for m in init.layers]
    train = data.train_mask.nonzero().view(-1)
    gen = torch.Generator().manual_seed(settings.SEED)
    nb = train.numel() // B
    assert nb * B == train.numel()
    batches, counters = [], []
    for epoch in range(1, epochs + 1):
        order = train[torch.randperm(train.numel(), generator=gen)]
        for b in range(nb):
            batches.append(order[b * B:(b + 1) * B])
            counters.append(batch_counter(epoch, nb, b, 0, 1))
    _, params = _oracle_train(data, layers, batches, fan, settings.SEED, counters, fig2a.LEARNING_RATE, B)
    # fig2a.test() walks the test loader twice per epoch (inference, then the labels): the last
    # inference pass is the loader's epoch 2 * (epochs - 1)
    test_ids = data.test_mask.nonzero().view(-1)
    nb_test = (test_ids.numel() + B - 1) // B
    acc_ref = _oracle_accuracy(data, params, test_ids, fan, settings.SEED, batch_counter(2 * (epochs - 1), nb_test, 0, 0, 1), B)
    assert abs(out['test_acc'] - acc_ref) <= 0.005, (out['test_acc'], acc_ref)


def test_fig2b_reported_f1_matches_oracle(dev):
    """f-4: what experiments/fig2b reports for one sample size (test micro-F1 of the trainer's
    step: cross entropy + unsupervised term, three forwards; reference experiments/fig2b/runners/
    runners.py:20-57) against the CPU oracle run on the same batches, positives and negatives."""
    from graphsage_amd.dist import batch_counter
    from graphsage_amd.experiments import fig2b
    from graphsage_amd import settings
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.trainers import dataloader_kwargs
    ds = _dataset(noise=1.5)
    data = ds[0]
    s, B, hidden, lr = 5, 128, 32, 1e-2
    old = dict(dataloader_kwargs)
    dataloader_kwargs['batch_size'] = B
    try:
        torch.manual_seed(5)
        out = fig2b.get(s, dataset=ds, num_epochs=1, num_runs=1, hidden_channels=hidden, learning_rate=lr).run()
    finally:
        dataloader_kwargs.update(old)
    torch.manual_seed(5)
    init = GraphSAGE(ds.num_features, hidden, "mean", 2, ds.num_classes)
    params = [{"lin_l": m.lin_l.weight.detach().clone(), "lin_r": m.lin_r.weight.detach().clone()} for m in init.layers]
    state = [{k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in p.items()} for p in params]
    c0, r0, _ = csc.to_csc(data.edge_index.numpy(), data.num_nodes)
    train = data.train_mask.nonzero().view(-1)
    nb = train.numel() // B
    seed = settings.SEED
    for b in range(nb):                                   # trainer loaders: shuffle=False, epoch 1
        seeds, counter = train[b * B:(b + 1) * B], batch_counter(1, nb, b, 0, 1)
        node, src, dst, _, _, _ = sampler.neighbor_sample(c0, r0, seeds.numpy(), [s, s], seed, counter)
        pos = sampler.random_neighbor(c0, r0, node, seed, counter)
        neg = sampler.random_nodes(data.num_nodes, len(node), seed, counter)
        leaf = [{k: v.clone().requires_grad_(True) for k, v in p.items()} for p in params]
        ei = torch.from_numpy(np.stack([src, dst]))
        outs = [sage_ref.graphsage_forward(data.x[torch.from_numpy(ids)], ei, leaf)[:B] for ids in (node, pos, neg)]
        loss = sage_ref.unsup_loss(*outs) + F.cross_entropy(outs[0], data.y[seeds])
        loss.backward()
        for p, lf, st in zip(params, leaf, state):
            for k in p:
                p[k], m1, v1 = sage_ref.adam_ref(p[k], lf[k].grad, st[k][0], st[k][1], b + 1, lr)
                st[k] = (m1, v1)
    test_ids = data.test_mask.nonzero().view(-1)
    f1_ref = _oracle_accuracy(data, params, test_ids, [s, s], seed, 0, B)     # test loader's first pass: epoch 0
    assert abs(out['test_f1'] - f1_ref) <= 0.005, (out['test_f1'], f1_ref)


def test_engine_evaluation_is_what_fig2a_and_the_trainers_call(dev, monkeypatch):
    """fig2a.test() and the supervised trainer's eval() run the engine's forward-only pipeline over
    the loader's seeds (engine.evaluate_loader) — not the module path — whenever they hold an
    engine; the loader's epoch advances by the two passes the reference makes per evaluation
    (inference, then labels: trainers/node_level_graphsage_trainers.py:54-55)."""
    import torch.nn.functional as F
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd.experiments import fig2a
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import UniformLoader
    from graphsage_amd.trainers import SupervisedGraphSageTrainerForNodeLevelTask, dataloader_kwargs
    calls = []
    real = SageTrainEngine.evaluate_loader

    def spy(self, loader, return_pred=False):
        out = real(self, loader, return_pred)
        calls.append(out is not None)
        return out
    monkeypatch.setattr(SageTrainEngine, "evaluate_loader", spy)
    monkeypatch.setattr(GraphSAGE, "inference", lambda self, loader: (_ for _ in ()).throw(AssertionError("module path taken")))
    ds = _dataset(noise=1.0)
    out = fig2a.get("mean", dataset=ds, num_epochs=2, batch_size=128, k1=6, k2=4, verbose=False).run()
    assert out['engine'] and calls == [True, True] and out['test_acc'] > 0.5
    calls.clear()
    data = ds[0]
    torch.manual_seed(0)
    model = GraphSAGE(ds.num_features, 32, "mean", 2, ds.num_classes).to(dev)
    monkeypatch.setitem(dataloader_kwargs, 'batch_size', 128)
    tr = SupervisedGraphSageTrainerForNodeLevelTask(data, UniformLoader, supervised_only=True, loss_fn=F.cross_entropy,
                                                    model=model, optimizer=torch.optim.Adam(model.parameters(), lr=1e-2),
                                                    num_epochs=1, device=dev, k1=6, k2=4)
    assert tr.engine is not None
    e0 = tr.test_loader.epoch
    res = tr.test()
    assert calls and all(calls) and tr.test_loader.epoch == e0 + 2
    assert 0.0 <= res['test_f1'] <= 1.0 and res['test_acc'] == res['test_f1']
