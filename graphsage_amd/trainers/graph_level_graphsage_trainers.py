"""Graph-level (PPI) GraphSAGE trainers — the reference's class names, constructor keywords and
result keys (graphsage/trainers/graph_level_graphsage_trainers.py:11-69, :72-159), written from
that interface.

  SupervisedGraphSageTrainerForGraphLevelTask    full-graph training (no sampler, :23-44): every
      batch is one or more whole graphs through GraphSAGE.forward; multi-label targets, metric =
      micro-F1 of (logit > 0) against y over all nodes and labels.
  UnsupervisedGraphSageTrainerForGraphLevelTask  one UniformLoader per training graph (:85-92),
      the random-walk loss on every sampled batch (:104-117), then a one-vs-rest logistic head on
      the frozen full-graph embeddings.  The reference fits sklearn's MultiOutputClassifier(
      SGDClassifier(loss="log")) there (:126-131); sklearn baselines are out of scope,
      the head is fitted on the device and plays the same role.

Micro-F1 is computed on the device from TP / FP / FN counts (2TP / (2TP + FP + FN)), which is what
sklearn.metrics.f1_score(average='micro') returns for multi-label indicator matrices.
"""
import torch
import torch.nn.functional as F

from .. import settings
from .base_trainers import GraphSageBaseTrainer, SupervisedGraphSageBaseTrainer, dataloader_kwargs


def micro_f1_multilabel(pred, target):
    """pred, target: {0,1} tensors of one shape -> micro-averaged F1 as a python float."""
    pred, target = pred.bool(), target.bool()
    tp = (pred & target).sum().double()
    fp = (pred & ~target).sum().double()
    fn = (~pred & target).sum().double()
    denom = 2 * tp + fp + fn
    return float(2 * tp / denom) if float(denom) > 0 else 0.0


class SupervisedGraphSageTrainerForGraphLevelTask(SupervisedGraphSageBaseTrainer):
    def __init__(self, train_loader, val_loader, test_loader, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.train_loader, self.val_loader, self.test_loader = train_loader, val_loader, test_loader

    def train(self, epoch):
        self.model.train()
        weighted = torch.zeros((), device=self.device)
        for graphs in self.train_loader:
            graphs = graphs.to(self.device)
            self.optimizer.zero_grad()
            loss = self.loss_fn(self.model(graphs.x, graphs.edge_index), graphs.y)
            loss.backward()
            self.optimizer.step()
            weighted += loss.detach() * graphs.num_graphs
        return weighted / len(self.train_loader.dataset)

    @torch.no_grad()
    def eval(self, loader):
        self.model.eval()
        preds, targets = [], []
        for graphs in loader:
            graphs = graphs.to(self.device)
            preds.append(self.model(graphs.x, graphs.edge_index) > 0)
            targets.append(graphs.y > 0.5)
        return micro_f1_multilabel(torch.cat(preds), torch.cat(targets))

    @torch.no_grad()
    def test(self):
        return {'train_f1': None if settings.NO_EVAL_TRAIN else self.eval(self.train_loader),
                'val_f1': self.eval(self.val_loader), 'test_f1': self.eval(self.test_loader)}


class UnsupervisedGraphSageTrainerForGraphLevelTask(GraphSageBaseTrainer):
    def __init__(self, train_loader, val_loader, loader, test_loader, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.train_loader, self.val_loader, self.test_loader = train_loader, val_loader, test_loader
        self.train_data_list = [graph for graph in train_loader.dataset]
        # every node of every training graph is a seed (input_nodes=None), reshuffled per epoch
        self.train_loader_list = [loader(graph, input_nodes=None, num_neighbors=[self.k1, self.k2], shuffle=True,
                                         **dataloader_kwargs) for graph in self.train_data_list]

    def train(self, epoch):
        self.model.train()
        weighted = torch.zeros((), device=self.device)
        nodes = 0
        for graph, sampled in zip(self.train_data_list, self.train_loader_list):
            nodes += graph.num_nodes
            for batch in sampled:
                batch = batch.to(self.device)
                self.optimizer.zero_grad()
                z = self.model(batch.x, batch.edge_index)[:batch.batch_size]
                loss = self.unsup_loss_fn(z, batch, graph)
                loss.backward()
                self.optimizer.step()
                weighted += loss.detach() * z.size(0)
        return weighted / nodes

    @torch.no_grad()
    def _loader_to_embeddings_and_labels(self, loader):
        zs, ys = [], []
        for graphs in loader:
            graphs = graphs.to(self.device)
            zs.append(self.model(graphs.x, graphs.edge_index))
            ys.append(graphs.y)
        return torch.cat(zs), torch.cat(ys)

    @staticmethod
    def _fit_one_vs_rest(z, y, epochs=300, lr=0.1):
        """Independent logistic regressions, one per label column, full batch on the device."""
        w = torch.zeros(z.size(1), y.size(1), device=z.device, requires_grad=True)
        b = torch.zeros(y.size(1), device=z.device, requires_grad=True)
        opt = torch.optim.Adam([w, b], lr=lr)
        target = y.float()
        with torch.enable_grad():
            for _ in range(epochs):
                opt.zero_grad()
                F.binary_cross_entropy_with_logits(z @ w + b, target).backward()
                opt.step()
        return w.detach(), b.detach()

    @torch.no_grad()
    def test(self):
        self.model.eval()
        z_train, y_train = self._loader_to_embeddings_and_labels(self.train_loader)
        w, b = self._fit_one_vs_rest(z_train, y_train)
        scores = {'train_f1': None if settings.NO_EVAL_TRAIN
                  else micro_f1_multilabel(z_train @ w + b > 0, y_train > 0.5)}
        for split, loader in (('val', self.val_loader), ('test', self.test_loader)):
            z, y = self._loader_to_embeddings_and_labels(loader)
            scores[f'{split}_f1'] = micro_f1_multilabel(z @ w + b > 0, y > 0.5)
        return scores
