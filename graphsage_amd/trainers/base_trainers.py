"""Trainer base classes: the callers on either side of the hot path.

Written from the interface the reference's scripts use (graphsage/trainers/base_trainers.py:
class names, constructor keywords, the per-epoch result keys and what run() returns), not from
its code.  Design of this side:

  * an epoch's loss stays a device tensor until the epoch ends — one host synchronisation per
    epoch, where the reference reads loss.item() every step;
  * the per-epoch records live in an EpochLog that knows how to pick the best epoch, print a
    line and write itself as JSON (the reference also draws PNGs with pandas + matplotlib:
    plotting is out of scope, §8);
  * nothing is created on disk unless run(save=True).
"""
import json
import os
import time
from datetime import datetime
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .. import settings
from ..samplers import get_pos_neg_batches

# loader keyword arguments shared by every trainer (the reference's module-level dict of the same
# name; `kwargs` is its second name there and some scripts import that one)
dataloader_kwargs = dict(batch_size=settings.BATCH_SIZE, num_workers=settings.NUM_WORKERS,
                         persistent_workers=settings.PERSISTENT_WORKERS)
kwargs = dataloader_kwargs


def capitalize(underscore_string: str) -> str:
    """'val_f1' -> 'Val F1' style labels for the epoch line."""
    return ' '.join(part[:1].upper() + part[1:].lower() for part in underscore_string.split('_'))


class EpochLog:
    """Per-epoch result records (dicts of floats), 1-based epochs."""

    def __init__(self):
        self.records: List[Dict[str, float]] = []

    def add(self, **values) -> Dict[str, float]:
        record = {key: value for key, value in values.items() if value is not None}
        self.records.append(record)
        return record

    def best(self, key: str) -> Tuple[int, Dict[str, float]]:
        """(epoch, record) with the largest `key`; records without it rank lowest, ties go to
        the earliest epoch."""
        if not self.records:
            raise RuntimeError("no epoch has been run")
        scores = [record.get(key, float('-inf')) for record in self.records]
        epoch = scores.index(max(scores))
        return epoch + 1, self.records[epoch]

    @staticmethod
    def line(epoch: int, record: Dict[str, float]) -> str:
        fields = [f'{capitalize(key)}: {value:.4f}' for key, value in record.items()]
        return f'Epoch: {epoch:02d}, ' + ', '.join(fields)

    def dump(self, directory: str, params: Dict[str, Any]) -> None:
        os.makedirs(directory)
        for name, payload in (('params.json', params), ('results.json', self.records)):
            with open(os.path.join(directory, name), 'w') as handle:
                json.dump(payload, handle)


class BaseTrainer:
    """Interface: train(epoch) -> loss, test() -> dict of metrics, run() -> best epoch's dict."""

    def __init__(self, dataset_name: Optional[str] = None):
        self.dataset_name = dataset_name

    def train(self, *args, **kwargs):
        raise NotImplementedError

    def test(self):
        raise NotImplementedError

    def run(self, *args, **kwargs):
        raise NotImplementedError


class TorchModuleBaseTrainer(BaseTrainer):
    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, num_epochs: int,
                 device: Any, *args, num_prints: int = 10, **kwargs):
        super().__init__(*args, **kwargs)
        self.model, self.optimizer = model, optimizer
        self.num_epochs, self.num_prints = int(num_epochs), int(num_prints)
        self.device = device
        self.log = EpochLog()

    @property
    def results(self) -> List[Dict[str, float]]:
        return self.log.records

    # ------------------------------------------------------------------ the epoch loop
    def _timed(self, fn, *args):
        """fn(*args) and its wall time; the device is drained on both sides, so the time is the
        work's, not its launch's."""
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        begin = time.perf_counter()
        out = fn(*args)
        if isinstance(out, torch.Tensor):
            out = float(out)                 # the epoch's one read-back of the loss
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        return out, time.perf_counter() - begin

    def run(self, return_best_epoch_only: bool = True, val_metric: str = 'f1', save: bool = False,
            verbose: bool = True):
        every = max(1, self.num_epochs // max(1, min(self.num_epochs, self.num_prints)))
        for epoch in range(1, self.num_epochs + 1):
            loss, train_time = self._timed(self.train, epoch)
            metrics, test_time = self._timed(self.test)
            record = self.log.add(loss=loss, **metrics, train_time=train_time, test_time=test_time)
            if verbose and epoch % every == 0:
                self.print_epoch_with_results(epoch, record)
        best_epoch, best = self.get_best_epoch_with_results(val_metric)
        if verbose:
            print('*** BEST ***')
            self.print_epoch_with_results(best_epoch, best)
        if save:
            self.save_results()
        return best if return_best_epoch_only else (best, self.results)

    @staticmethod
    def print_epoch_with_results(epoch, epoch_results):
        print(EpochLog.line(epoch, epoch_results))

    def get_best_epoch_with_results(self, val_metric):
        return self.log.best(f'val_{val_metric}')

    # ------------------------------------------------------------------ persistence
    def save_results(self) -> str:
        model_name = getattr(self.model, 'name', type(self.model).__name__)
        stamp = datetime.now().strftime('%Y-%m-%d_%H-%M-%S_%f')
        tag = f'{stamp}_dataset={self.dataset_name}_model={model_name}_num_epochs={self.num_epochs}'
        root = getattr(settings, 'RESULTS_DIR', os.path.join(settings.ROOT_DIR, 'results'))
        directory = os.path.join(root, 'trainers', tag)
        self.log.dump(directory, {
            'dataset': self.dataset_name, 'model': str(self.model), 'optimizer': str(self.optimizer),
            'num_epochs': self.num_epochs, 'device': torch.device(self.device).type})
        print(f'Results saved to {directory}')
        return directory


class SupervisedTorchModuleBaseTrainer(TorchModuleBaseTrainer):
    def __init__(self, loss_fn, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.loss_fn = loss_fn


class GraphSageBaseTrainer(TorchModuleBaseTrainer):
    """Holds the two fan-outs (25 / 10 by default) and the unsupervised loss."""

    def __init__(self, k1: int = 25, k2: int = 10, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.k1, self.k2 = k1, k2

    def unsup_loss_fn(self, out, batch, data, use_triple_loss=settings.args.use_triple_loss, counter=None):
        """The random-walk loss of the paper on a sampled batch: the model is run twice more over
        the batch's OWN edge_index, once with the features of one positive node per batch node
        (a uniformly chosen in-neighbour) and once with those of a uniform negative node; the
        seeds' three embeddings z, z+, z- give  -mean log s(z.z+) - mean log s(-z.z-)  — or the
        triplet margin loss when the --use_triple_loss flag is set."""
        seeds = batch.batch_size
        z_pos, z_neg = (self.model(other.x, other.edge_index)[:seeds]
                        for other in get_pos_neg_batches(batch, data, counter=counter))
        if use_triple_loss:
            return F.triplet_margin_loss(out, z_pos, z_neg)
        attract = F.logsigmoid(torch.sum(out * z_pos, dim=-1))
        repel = F.logsigmoid(-torch.sum(out * z_neg, dim=-1))
        return -(attract.mean() + repel.mean())


class SupervisedGraphSageBaseTrainer(GraphSageBaseTrainer, SupervisedTorchModuleBaseTrainer):
    pass
