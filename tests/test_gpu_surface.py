"""The reference's import surface (`graphsage.*`, and the top-level `models` / `samplers` /
`datasets` / `layers` spellings its scripts use) served by compat/, driven the way
examples/citation_graphsage_supervised.py:1-36 drives it (the script itself cannot travel to the
GPU box, so the same calls are spelled out here), and the graph-level (PPI) trainers
(reference trainers/graph_level_graphsage_trainers.py) against the oracle."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sage_ref
from tests.helpers import load_layers, make_layers

pytestmark = pytest.mark.gpu
COMPAT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "compat")
TOP = ("graphsage", "models", "samplers", "datasets", "layers")


@pytest.fixture
def compat_path():
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in TOP}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, COMPAT)
    yield
    sys.path.remove(COMPAT)
    for k in [k for k in sys.modules if k.split(".")[0] in TOP]:
        del sys.modules[k]
    sys.modules.update(saved)


def test_reference_import_surface_runs_the_citation_example(dev, compat_path):
    import os.path as osp
    from graphsage import settings
    from graphsage.datasets import Planetoid
    from graphsage.models import GraphSAGE
    from graphsage.samplers import UniformLoader
    from graphsage.trainers import SupervisedGraphSageTrainerForNodeLevelTask
    from models.graphsage import GraphSAGE as GraphSAGE2           # examples/reddit_graphsage_supervised.py:9
    from samplers import get_pos_neg_batches                       # trainers/base_trainers.py:15
    from layers import SAGE
    import graphsage_amd
    assert GraphSAGE2 is GraphSAGE is graphsage_amd.models.GraphSAGE and SAGE is graphsage_amd.layers.SAGE
    assert get_pos_neg_batches is graphsage_amd.samplers.get_pos_neg_batches
    device = settings.DEVICE
    dataset_name = 'Cora'
    path = osp.join(settings.DATA_DIR, dataset_name)
    dataset = Planetoid(path, dataset_name)
    data = dataset[0].to(device, 'x', 'y')
    torch.manual_seed(0)
    model = GraphSAGE(in_channels=dataset.num_features, hidden_channels=256, out_channels=dataset.num_classes,
                      num_layers=2, aggregator=settings.args.aggregator).to(device)
    best = SupervisedGraphSageTrainerForNodeLevelTask(
        dataset_name=dataset_name, model=model, data=data, loader=UniformLoader, num_epochs=3,
        loss_fn=torch.nn.CrossEntropyLoss(), optimizer=torch.optim.Adam(model.parameters(), lr=0.01),
        device=device).run(verbose=False)
    assert set(best) >= {'loss', 'train_f1', 'val_f1', 'test_f1', 'train_time', 'test_time'}
    assert best['test_f1'] > 0.3 and best['loss'] == best['loss']
    assert not osp.exists(path)                                    # nothing is created on disk


def _ppi_sets():
    from graphsage_amd.data import GraphDataLoader
    from graphsage_amd.datasets import PPI
    train, val, test = PPI(split='train'), PPI(split='val'), PPI(split='test')
    small = type(train)(train._datas[:3], train.num_classes)      # three training graphs keep the oracle fast
    return small, val, test, GraphDataLoader


def test_graph_level_supervised_trainer_matches_oracle(dev):
    """Full-graph PPI-style training (reference graph_level_graphsage_trainers.py:23-44): one epoch
    of per-graph BCE steps + Adam against the oracle (same order: shuffle=False), and the reported
    micro-F1 against sklearn's f1_score on the oracle's predictions."""
    from sklearn.metrics import f1_score
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.trainers import SupervisedGraphSageTrainerForGraphLevelTask
    train, val, test, Loader = _ppi_sets()
    layers = make_layers([50, 64, 121], "mean", 2)
    model = GraphSAGE(50, 64, "mean", 2, 121).to(dev)
    load_layers(model, layers)
    lr = 5e-3
    tr = SupervisedGraphSageTrainerForGraphLevelTask(
        dataset_name='PPI', model=model, loss_fn=torch.nn.BCEWithLogitsLoss(),
        optimizer=torch.optim.Adam(model.parameters(), lr=lr),
        train_loader=Loader(train, batch_size=1, shuffle=False), val_loader=Loader(val, batch_size=2),
        test_loader=Loader(test, batch_size=2), device=dev, num_epochs=1)
    best = tr.run(verbose=False)
    params = [{k: v.clone() for k, v in p.items()} for p in layers]
    state = [{k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in p.items()} for p in params]
    total = 0.0
    for step, g in enumerate(train, start=1):
        leaf = [{k: v.clone().requires_grad_(True) for k, v in p.items()} for p in params]
        loss = F.binary_cross_entropy_with_logits(sage_ref.graphsage_forward(g.x, g.edge_index, leaf), g.y)
        loss.backward()
        total += float(loss)
        for p, lf, st in zip(params, leaf, state):
            for k in p:
                p[k], m1, v1 = sage_ref.adam_ref(p[k], lf[k].grad, st[k][0], st[k][1], step, lr)
                st[k] = (m1, v1)
    assert abs(best['loss'] - total / len(train)) <= 1e-4 * max(1.0, total / len(train))
    for m, p in zip(model.layers, params):
        for k in p:
            assert float((getattr(m, k).weight.detach().cpu() - p[k]).norm() / p[k].norm()) < 1e-3
    with torch.no_grad():
        ys = torch.cat([g.y for g in test]).numpy()
        preds = torch.cat([(sage_ref.graphsage_forward(g.x, g.edge_index, params) > 0).float() for g in test]).numpy()
    assert abs(best['test_f1'] - f1_score(ys, preds, average='micro')) <= 0.005
    assert set(best) >= {'loss', 'train_f1', 'val_f1', 'test_f1'}


def test_graph_level_unsupervised_trainer_runs(dev):
    """One UniformLoader per training graph + the random-walk loss (reference :85-117); the
    embeddings must carry the planted structure (micro-F1 of the one-vs-rest head above the
    all-negative baseline)."""
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import UniformLoader
    from graphsage_amd.trainers import UnsupervisedGraphSageTrainerForGraphLevelTask, dataloader_kwargs
    train, val, test, Loader = _ppi_sets()
    torch.manual_seed(1)
    model = GraphSAGE(50, 64, "mean", 2).to(dev)
    old = dict(dataloader_kwargs)
    dataloader_kwargs['batch_size'] = 512
    try:
        tr = UnsupervisedGraphSageTrainerForGraphLevelTask(
            dataset_name='PPI', model=model, optimizer=torch.optim.Adam(model.parameters(), lr=5e-3),
            train_loader=Loader(train, batch_size=1), val_loader=Loader(val, batch_size=2), loader=UniformLoader,
            test_loader=Loader(test, batch_size=2), device=dev, num_epochs=2)
    finally:
        dataloader_kwargs.update(old)
    assert len(tr.train_loader_list) == 3
    best = tr.run(verbose=False)
    assert best['loss'] > 0 and 0.4 < best['test_f1'] <= 1.0 and 'train_f1' in best
