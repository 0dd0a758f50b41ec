"""fig2a-style timing harness (reference experiments/fig2a/runners/graphsage_runners.py:30-116):
per aggregator, the average time of one training batch (sample -> forward -> cross entropy ->
backward -> Adam, supervised term only, :38-46) and of one inference pass over the test split
(:57-66), plus the test accuracy the reference prints per epoch (:92).

Same function names and return keys ('train_time' per batch, 'test_time' per full test pass).
Timing differs on purpose: the reference wraps every batch in time.time() (:39,:47), which on a
GPU measures launch time unless it synchronises; here the device is synchronised once per epoch
and the epoch time is divided by the number of batches.  mean/sum/max models run on the static-shape
engine; every other aggregator runs the module path (autograd Functions over the same kernels).

    python -m graphsage_amd.experiments.fig2a --dataset cora --aggregators mean max_pool
"""
import argparse
import os
import json
import time

import torch
import torch.nn.functional as F

from .. import settings
from ..datasets import Planetoid, Reddit
from ..dist import all_reduce_gradients
from ..models import GraphSAGE
from ..samplers import UniformLoader

K1, K2 = 25, 10                    # fig2a/settings.py (reference): fan-outs of the sweep
HIDDEN_CHANNELS, NUM_LAYERS, LEARNING_RATE, BATCH_SIZE = 256, 2, 1e-3, 512


def load_dataset(name):
    name = name.lower()
    if name == 'reddit-shaped':
        # timing runs: Reddit's SHAPE generated on the device (random features and labels: the
        # accuracy it reports is chance level by construction), with Reddit's split sizes
        from ..datasets.synthetic import SHAPES, SyntheticDataset, _masks, reddit_like_device
        n, e, f, c, tr, va, te = SHAPES['Reddit']
        data = reddit_like_device(n, e, f, c, seed=0, device=settings.DEVICE)
        _masks(data, tr, va, te, 0)
        return SyntheticDataset([data], c)
    if name == 'reddit':
        return Reddit(None)
    if name in {'cora', 'citeseer', 'pubmed'}:
        return Planetoid(None, name.capitalize())
    raise ValueError(f"unknown dataset {name}")


def train(epoch, model, optimizer, train_loader, engine=None):
    """Returns (mean loss, seconds per batch)."""
    model.train()
    dev = settings.DEVICE
    torch.cuda.synchronize()
    start = time.time()
    if engine is not None:
        # the loader's own order for this epoch: a fresh permutation when it was built with
        # shuffle=True (fig2a/runners/graphsage_runners.py:72-73), partial last batch included
        seeds = train_loader._order()
        train_loader.epoch += 1
        loss = engine.train_epoch(seeds, epoch=epoch, drop_last=False)
        nb = len(train_loader)
        torch.cuda.synchronize()
        return float(loss), (time.time() - start) / nb
    total = torch.zeros((), device=dev)
    examples = 0
    for batch in train_loader:
        optimizer.zero_grad()
        y = batch.y[:batch.batch_size].to(dev)
        y_hat = model(batch.x.to(dev), batch.edge_index.to(dev))[:batch.batch_size]
        loss = F.cross_entropy(y_hat, y)
        loss.backward()
        all_reduce_gradients(model)      # data parallel: average over ranks (no-op otherwise)
        optimizer.step()
        total += loss.detach() * batch.batch_size
        examples += batch.batch_size
    torch.cuda.synchronize()
    return float(total) / examples, (time.time() - start) / len(train_loader)


@torch.no_grad()
def test(model, subgraph_loader, engine=None):
    """Returns (accuracy, seconds for the full pass).  With an engine the pass runs on its
    forward-only pipeline (engine.evaluate_loader: the same sampled neighbourhoods, arg-max on
    the device); the timed part is the same — everything up to the predictions."""
    torch.cuda.synchronize()
    start = time.time()
    if engine is not None:
        model.eval()
        out = engine.evaluate_loader(subgraph_loader)
        if out is not None:
            torch.cuda.synchronize()
            test_time = time.time() - start
            return int(out[0]) / int(out[1]), test_time
    y_hat = model.inference(subgraph_loader).argmax(dim=-1)
    torch.cuda.synchronize()
    test_time = time.time() - start
    y = torch.cat([batch.y[:batch.batch_size] for batch in subgraph_loader]).to(y_hat.device)
    return int((y_hat == y).sum()) / int(y.shape[0]), test_time


def run(aggregator, dataset=None, num_epochs=None, batch_size=BATCH_SIZE, k1=K1, k2=K2, use_engine=None,
        verbose=True):
    dataset = dataset if dataset is not None else load_dataset('cora')
    num_epochs = num_epochs or settings.NUM_EPOCHS
    data = dataset[0]
    kwargs = {'batch_size': batch_size, 'num_workers': settings.NUM_WORKERS}
    train_loader = UniformLoader(data, input_nodes=data.train_mask, num_neighbors=[k1, k2], shuffle=True, **kwargs)
    test_loader = UniformLoader(data, input_nodes=data.test_mask, num_neighbors=[k1, k2], shuffle=False, **kwargs)
    model = GraphSAGE(in_channels=dataset.num_features, hidden_channels=HIDDEN_CHANNELS,
                      out_channels=dataset.num_classes, num_layers=NUM_LAYERS,
                      aggregator=aggregator).to(settings.DEVICE)
    optimizer = torch.optim.Adam(model.parameters(), lr=LEARNING_RATE)
    engine = None
    n_train = train_loader._idx_dev.numel()
    if use_engine is None:
        use_engine = aggregator in ('mean', 'sum', 'max', 'gcn', 'max_pool', 'mean_pool', 'lstm') and n_train >= batch_size
    if use_engine:
        from ..engine import SageTrainEngine
        graph = train_loader.neighbor_sampler.graph
        engine = SageTrainEngine(model, train_loader.neighbor_sampler, graph.features(data), graph.labels(data),
                                 batch_size, lr=LEARNING_RATE,
                                 overlap_gather=False if os.environ.get('SAGE_FIG2A_SERIAL') else 'auto')
        # (bench.py's headline schedule: the next batch's layer-0 gather beside this batch's step; through this
        # harness, 4 epochs at the Reddit shape: mean 0.269-0.278 against 0.282-0.287 ms per batch on one stream,
        # max 0.289 against 0.302.  The pooling engine ignores it.)
    train_time = test_time = 0.0
    loss = test_acc = None
    for epoch in range(1, num_epochs + 1):
        loss, epoch_train_time = train(epoch, model, optimizer, train_loader, engine)
        test_acc, epoch_test_time = test(model, test_loader, engine)
        if verbose:
            print(f'Epoch: {epoch:02d}, Loss: {loss:.4f}, Test Acc: {test_acc:.4f}')
        train_time += epoch_train_time
        test_time += epoch_test_time
    # (the reference's keys are the averages over the epochs, graphsage_runners.py:100-103; the last
    # epoch's own times are reported beside them: the first epoch of a process carries one-off
    # costs — code-object loading, call-tape recording — that an average over two epochs still shows)
    return {'test_time': test_time / num_epochs, 'train_time': train_time / num_epochs,
            'test_acc': test_acc, 'loss': loss, 'engine': engine is not None,
            'test_time_last': epoch_test_time, 'train_time_last': epoch_train_time}


class GraphSAGERunner:
    def __init__(self, aggregator, **kw):
        self.aggregator = aggregator
        self.kw = kw

    def run(self):
        return run(self.aggregator, **self.kw)


def get(aggregator, **kw):
    return GraphSAGERunner(aggregator, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dataset', default='cora')
    ap.add_argument('--aggregators', nargs='+', default=['mean', 'max_pool'])
    ap.add_argument('--epochs', type=int, default=2)
    ap.add_argument('--module-path', action='store_true', help='never use the static-shape engine')
    a, _ = ap.parse_known_args()
    ds = load_dataset(a.dataset)
    for agg in a.aggregators:
        kw = {'use_engine': False} if a.module_path else {}
        print(json.dumps({'aggregator': agg, **get(agg, dataset=ds, num_epochs=a.epochs, verbose=False, **kw).run()}),
              flush=True)


if __name__ == '__main__':
    main()
