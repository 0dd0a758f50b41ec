"""Epoch loop, result bookkeeping and the unsupervised loss — the callers on either side of the
hot path.  Same classes, constructor arguments and result keys as the
reference's graphsage/trainers/base_trainers.py:27-185; what differs:

  * nothing here synchronises per step: losses stay device tensors until the epoch ends;
  * save_results writes params.json / results.json only (the reference also draws one PNG per
    column with pandas + matplotlib, base_trainers.py:84-95 — plotting is out of scope, §8);
  * RESULTS_DIR defaults to <repo>/results and is only created when run(save=True).
"""
import json
import os
import os.path as osp
import time
from datetime import datetime as dt
from typing import Any

import torch
import torch.nn.functional as F
from torch import nn
from torch.optim import Optimizer

from .. import settings
from ..samplers import get_pos_neg_batches

dataloader_kwargs = kwargs = {
    'batch_size': settings.BATCH_SIZE,
    'num_workers': settings.NUM_WORKERS,
    'persistent_workers': settings.PERSISTENT_WORKERS,
}


def capitalize(underscore_string):
    return ' '.join(w.capitalize() for w in underscore_string.split('_'))


class BaseTrainer:
    def __init__(self, dataset_name: str = None):
        self.dataset_name = dataset_name

    def train(self, *args, **kwargs) -> float:
        raise NotImplementedError

    def test(self) -> dict:
        raise NotImplementedError

    def run(self, *args, **kwargs) -> dict:
        raise NotImplementedError


class TorchModuleBaseTrainer(BaseTrainer):
    def __init__(self, model: nn.Module, optimizer: Optimizer, num_epochs: int, device: Any, *args,
                 num_prints: int = 10, **kwargs):
        super().__init__(*args, **kwargs)
        self.model = model
        self.optimizer = optimizer
        self.num_epochs = num_epochs
        self.num_prints = num_prints
        self.device = device
        self.results = []

    def save_results(self):
        name = self.model.name if hasattr(self.model, 'name') else self.model.__class__.__name__
        folder = {'dataset': self.dataset_name, 'model': name, 'num_epochs': self.num_epochs}
        params = {'dataset': self.dataset_name, 'model': str(self.model), 'optimizer': str(self.optimizer),
                  'num_epochs': self.num_epochs, 'device': torch.device(self.device).type}
        date = str(dt.now()).replace(' ', '_').replace(':', '-').replace('.', '_')
        folder_name = '_'.join([date] + [f'{k}={v}' for k, v in folder.items()])
        results_dir = getattr(settings, 'RESULTS_DIR', osp.join(settings.ROOT_DIR, 'results'))
        results_path = osp.join(results_dir, 'trainers', folder_name)
        os.makedirs(results_path)
        with open(osp.join(results_path, 'params.json'), 'w') as f:
            json.dump(params, f)
        with open(osp.join(results_path, 'results.json'), 'w') as f:
            json.dump(self.results, f)
        print(f'Results saved to {results_path}')
        return results_path

    def run(self, return_best_epoch_only=True, val_metric='f1', save=False, verbose=True):
        for epoch in range(1, self.num_epochs + 1):
            start = time.time()
            loss = self.train(epoch)
            loss = float(loss)          # the one host synchronisation of the epoch
            train_time = time.time() - start
            start = time.time()
            test_results = self.test()
            test_time = time.time() - start
            epoch_results = {'loss': loss, **test_results, 'train_time': train_time, 'test_time': test_time}
            epoch_results = {k: v for k, v in epoch_results.items() if v is not None}
            if verbose and epoch % (self.num_epochs // min(self.num_epochs, self.num_prints)) == 0:
                self.print_epoch_with_results(epoch, epoch_results)
            self.results.append(epoch_results)
        best_epoch, best_results = self.get_best_epoch_with_results(val_metric)
        if verbose:
            print('*** BEST ***')
            self.print_epoch_with_results(best_epoch, best_results)
        if save:
            self.save_results()
        if return_best_epoch_only:
            return best_results
        return best_results, self.results

    @staticmethod
    def print_epoch_with_results(epoch, epoch_results):
        txt = ', '.join([f'{capitalize(k)}: {v:.4f}' for k, v in epoch_results.items()])
        print(f'Epoch: {epoch:02d}, {txt}')

    def get_best_epoch_with_results(self, val_metric):
        best = max(self.results, key=lambda x: x.get(f'val_{val_metric}', 0))
        return self.results.index(best) + 1, best


class SupervisedTorchModuleBaseTrainer(TorchModuleBaseTrainer):
    def __init__(self, loss_fn, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.loss_fn = loss_fn


class GraphSageBaseTrainer(TorchModuleBaseTrainer):
    def __init__(self, k1: int = 25, k2: int = 10, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.k1 = k1
        self.k2 = k2

    def unsup_loss_fn(self, out, batch, data, use_triple_loss=settings.args.use_triple_loss):
        """reference base_trainers.py:170-181: two more forwards over the SAME edge_index with
        the positive / negative nodes' features in place of the batch's."""
        pos_batch, neg_batch = get_pos_neg_batches(batch, data)
        pos_out = self.model(pos_batch.x, pos_batch.edge_index)[:batch.batch_size]
        neg_out = self.model(neg_batch.x, neg_batch.edge_index)[:batch.batch_size]
        if use_triple_loss:
            return F.triplet_margin_loss(out, pos_out, neg_out)
        pos_loss = F.logsigmoid((out * pos_out).sum(-1)).mean()
        neg_loss = F.logsigmoid(-(out * neg_out).sum(-1)).mean()
        return -pos_loss - neg_loss


class SupervisedGraphSageBaseTrainer(GraphSageBaseTrainer, SupervisedTorchModuleBaseTrainer):
    pass
