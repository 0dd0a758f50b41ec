"""Node-level GraphSAGE trainers (reference graphsage/trainers/node_level_graphsage_trainers.py).

SupervisedGraphSageTrainerForNodeLevelTask.train is the reference's step (:34-47): forward on
the sampled batch, loss_fn on the seed rows PLUS the unsupervised term (two more forwards),
backward, optimizer step.  Differences: the loss is accumulated on the device (one sync per
epoch instead of loss.item() per step, :46); micro-F1 is computed on the device — for
single-label predictions f1_score(average='micro') equals accuracy, which is what :56-57 compute
twice.  With `supervised_only=True` (fig2a's step definition, experiments/fig2a/runners/
graphsage_runners.py:38-46) and a mean/sum model trained with cross entropy + Adam, the epoch runs
on the static-shape engine (graphsage_amd/engine.py); `use_engine=True` also puts the full step
(supervised + unsupervised term, three forwards) on the engine.

The unsupervised trainer's downstream classifier is sklearn's SGDClassifier in the reference
(:122-125); sklearn baselines are out of scope (§8), so test() here fits a logistic-regression
head on the device with the same role and reports the same keys.
"""
import torch
import torch.nn.functional as F

from .. import settings
from ..dist import all_reduce_gradients
from .base_trainers import GraphSageBaseTrainer, SupervisedGraphSageBaseTrainer, dataloader_kwargs


def _micro_f1(y_hat, y):
    """accuracy == micro-averaged F1 for single-label multi-class predictions."""
    acc = (y_hat == y).sum().float() / max(int(y.numel()), 1)
    return acc


def _dp_shard():
    """Loader kwargs that shard the seeds over the ranks of an initialised process group (one
    process per GPU; evaluation loaders stay whole on every rank)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return {'rank': dist.get_rank(), 'world_size': dist.get_world_size()}
    return {}


def _dp_engine():
    import torch.distributed as dist
    kw = _dp_shard()
    return {'world_size': kw['world_size'], 'process_group': dist.group.WORLD} if kw else {}


def _engine_refusal(trainer, unsup):
    """Why the static-shape engine cannot run this trainer's step (None if it can): it implements
    Adam, plain mean cross entropy on int64 labels, the logsigmoid unsupervised term, the
    aggregators of engine._REDUCE, and full batches of the loader's batch size."""
    from ..engine import _REDUCE
    if type(trainer.optimizer) is not torch.optim.Adam or len(trainer.optimizer.param_groups) != 1 \
            or trainer.optimizer.param_groups[0].get('amsgrad'):
        return "the engine implements plain Adam with one parameter group"
    if trainer.model.aggregator not in _REDUCE:
        return f"aggregator {trainer.model.aggregator!r} runs on the module path"
    if unsup and trainer.model.aggregator in ('max_pool', 'mean_pool', 'lstm'):
        return "the engine runs the pooling and lstm aggregators in the supervised (fig2a) step only"
    if unsup != 'only':
        fn = getattr(trainer, 'loss_fn', None)
        plain = fn is F.cross_entropy or (isinstance(fn, torch.nn.CrossEntropyLoss) and fn.weight is None
                                          and fn.reduction == 'mean' and fn.label_smoothing == 0.0
                                          and fn.ignore_index == -100)
        if not plain:
            return "the engine's supervised term is plain mean cross entropy"
        y = getattr(trainer.data, 'y', None)
        if y is None or y.dtype != torch.int64 or y.dim() != 1:
            return "the engine needs single-label int64 targets"
    if unsup and settings.args.use_triple_loss:
        return "the engine's unsupervised term is the logsigmoid loss (--use_triple_loss is set)"
    if len(trainer.train_loader.dataset) < trainer.train_loader.batch_size:
        return "fewer training seeds than one batch"
    return None


def _engine_for(trainer, unsup, explicit):
    """Static-shape engine over the trainer's model / optimizer settings / train loader; None when
    the step is outside what the engine implements and it was not asked for explicitly."""
    from ..engine import SageTrainEngine
    why = _engine_refusal(trainer, unsup)
    if why is not None:
        if explicit:
            raise RuntimeError(f"use_engine=True: {why}")
        return None
    g = trainer.optimizer.param_groups[0]
    sampler = trainer.train_loader.neighbor_sampler
    graph = sampler.graph
    return SageTrainEngine(trainer.model, sampler, graph.features(trainer.data), graph.labels(trainer.data),
                           trainer.train_loader.batch_size, lr=g['lr'], betas=g['betas'], eps=g['eps'],
                           weight_decay=g['weight_decay'], unsup=unsup,
                           overlap_gather='auto',
                           **_dp_engine())


class SupervisedGraphSageTrainerForNodeLevelTask(SupervisedGraphSageBaseTrainer):
    def __init__(self, data, loader, *args, supervised_only=False, use_engine=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.data = data
        self.supervised_only = supervised_only
        mk = lambda mask, **kw: loader(data, input_nodes=mask, num_neighbors=[self.k1, self.k2], shuffle=False,
                                       **dataloader_kwargs, **kw)
        self.train_loader = mk(data.train_mask, **_dp_shard())   # training seeds sharded over ranks
        # evaluation on the training split sees ALL of it on every rank, like val / test
        self.train_eval_loader = mk(data.train_mask) if _dp_shard() else self.train_loader
        self.val_loader = mk(data.val_mask)
        self.test_loader = mk(data.test_mask)
        # use_engine=None: the engine where it implements exactly this trainer's step (fig2a's
        # step definition: supervised_only), else the module path; True: the engine or an error
        # 'auto': the engine whenever it implements this step, silently the module path otherwise
        self.engine = None
        if use_engine or (use_engine is None and supervised_only):
            self.engine = _engine_for(self, unsup=not self.supervised_only, explicit=use_engine is True)

    def train(self, epoch):
        self.model.train()
        if self.engine is not None:
            seeds = self.train_loader._idx_dev
            # (the loaders keep a partial last batch; the engine trains it padded and masked)
            return self.engine.train_epoch(seeds, epoch=epoch, rank=_dp_shard().get('rank', 0), drop_last=False)
        total = torch.zeros((), device=self.device)
        count = 0
        for batch in self.train_loader:
            batch = batch.to(self.device)
            self.optimizer.zero_grad()
            y = batch.y[:batch.batch_size]
            y_hat = self.model(batch.x, batch.edge_index)[:batch.batch_size]
            loss = self.loss_fn(y_hat, y)
            if not self.supervised_only:
                loss = loss + self.unsup_loss_fn(y_hat, batch, self.data)
            loss.backward()
            all_reduce_gradients(self.model)      # data parallel: average over ranks (no-op otherwise)
            self.optimizer.step()
            total += loss.detach() * batch.batch_size
            count += batch.batch_size
        return total / count

    @torch.no_grad()
    def eval(self, loader):
        if self.engine is not None and not self.engine.unsup:
            # the engine's forward-only pipeline over the same sampled neighbourhoods
            self.model.eval()
            out = self.engine.evaluate_loader(loader)
            if out is not None:
                acc = int(out[0]) / int(out[1])
                return acc, acc
        y_hat = self.model.inference(loader).argmax(dim=-1)
        y = torch.cat([batch.y[:batch.batch_size] for batch in loader]).to(y_hat.device)
        acc = float(_micro_f1(y_hat, y))
        return acc, acc

    @torch.no_grad()
    def test(self):
        train_acc, train_f1 = self.eval(self.train_eval_loader) if not settings.NO_EVAL_TRAIN else (None, None)
        val_acc, val_f1 = self.eval(self.val_loader)
        test_acc, test_f1 = self.eval(self.test_loader)
        return {'train_acc': train_acc, 'train_f1': train_f1, 'val_acc': val_acc, 'val_f1': val_f1,
                'test_acc': test_acc, 'test_f1': test_f1}


class UnsupervisedGraphSageTrainerForNodeLevelTask(GraphSageBaseTrainer):
    def __init__(self, data, loader, *args, use_engine=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.data = data
        mk = lambda mask, **kw: loader(data, input_nodes=mask, num_neighbors=[self.k1, self.k2], shuffle=False,
                                       **dataloader_kwargs, **kw)
        self.train_loader = mk(data.train_mask, **_dp_shard())   # training seeds sharded over ranks
        # the downstream head is fitted on the embeddings of the WHOLE training split on every rank
        self.train_eval_loader = mk(data.train_mask) if _dp_shard() else self.train_loader
        self.val_loader = mk(data.val_mask)
        self.test_loader = mk(data.test_mask)
        # use_engine: the epoch runs on the static-shape engine (loss = the unsupervised term
        # alone, three forwards per batch; mean / sum / max / gcn models with Adam)
        self.engine = _engine_for(self, unsup='only', explicit=True) if use_engine else None

    def train(self, epoch):
        self.model.train()
        if self.engine is not None:
            return self.engine.train_epoch(self.train_loader._idx_dev, epoch=epoch, rank=_dp_shard().get('rank', 0))
        total = torch.zeros((), device=self.device)
        count = 0
        for batch in self.train_loader:
            batch = batch.to(self.device)
            self.optimizer.zero_grad()
            out = self.model(batch.x, batch.edge_index)[:batch.batch_size]
            loss = self.unsup_loss_fn(out, batch, self.data)
            loss.backward()
            all_reduce_gradients(self.model)      # data parallel: average over ranks (no-op otherwise)
            self.optimizer.step()
            total += loss.detach() * out.size(0)
            count += batch.batch_size
        return total / count

    def _fit_head(self, x, y, num_classes, epochs=200, lr=0.5):
        """Multinomial logistic regression on frozen embeddings (role of the reference's
        SGDClassifier(loss='log'), :122-125), full-batch on the device."""
        w = torch.zeros(x.size(1), num_classes, device=x.device, requires_grad=True)
        b = torch.zeros(num_classes, device=x.device, requires_grad=True)
        opt = torch.optim.Adam([w, b], lr=lr)
        with torch.enable_grad():
            for _ in range(epochs):
                opt.zero_grad()
                F.cross_entropy(x @ w + b, y).backward()
                opt.step()
        return w.detach(), b.detach()

    @torch.no_grad()
    def test(self):
        y_all = self.data.y.to(self.device)
        masks = {k: getattr(self.data, f'{k}_mask').to(self.device) for k in ('train', 'val', 'test')}
        outs = {'train': self.model.inference(self.train_eval_loader)}
        num_classes = int(y_all.max()) + 1
        w, b = self._fit_head(outs['train'], y_all[masks['train']], num_classes)
        outs['val'] = self.model.inference(self.val_loader)
        outs['test'] = self.model.inference(self.test_loader)
        res = {}
        for k in ('train', 'val', 'test'):
            if k == 'train' and settings.NO_EVAL_TRAIN:
                res['train_acc'] = res['train_f1'] = None
                continue
            pred = (outs[k] @ w + b).argmax(-1)
            acc = float(_micro_f1(pred, y_all[masks[k]]))
            res[f'{k}_acc'] = res[f'{k}_f1'] = acc
        return res
