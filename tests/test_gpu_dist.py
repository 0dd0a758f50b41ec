"""Data-parallel engine path on the device: two ranks (two processes sharing the one GPU of the
test box, gloo carrying the gradient all-reduce in place of RCCL — RCCL refuses two ranks on one
device) run SageTrainEngine with world_size=2 on strided seed shards.  Checked: the replicas stay
bit-identical, and they follow a single-process emulation that computes both ranks' gradients,
averages them and applies one Adam step."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import assert_trajectories_close

pytestmark = pytest.mark.gpu

B, STEPS, FAN = 32, 4, [5, 3]
LONG_STEPS = 80


def _setup(aggregator="mean"):
    from graphsage_amd.datasets import planted_partition
    from graphsage_amd.models import GraphSAGE
    data = planted_partition(1500, 30000, 24, 5, seed=8)
    torch.manual_seed(3)
    model = GraphSAGE(24, 16, aggregator, 2, 5)
    train = torch.randperm(1500, generator=torch.Generator().manual_seed(1))[:2 * B * STEPS]
    return data, model, train


def _engine(data, model, dev, **kw):
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd.samplers import NeighborSampler
    from graphsage_amd.samplers.uniform_loader import device_graph
    graph = device_graph(data)
    return SageTrainEngine(model.to(dev), NeighborSampler(data, FAN, seed=5), graph.features(data),
                           graph.labels(data), B, lr=1e-2, **kw)


def _worker(rank, world, port, out, long_run=False, aggregator="mean"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from graphsage_amd.dist import batch_counter, shard_seeds
    dev = torch.device("cuda:0")
    data, model, train = _setup(aggregator)
    if long_run:      # bench.py's loop shape: many steps, no loss read-back in between
        train = torch.randint(0, 1500, (2 * B * LONG_STEPS,), generator=torch.Generator().manual_seed(2))
    eng = _engine(data, model, dev, world_size=world, process_group=dist.group.WORLD)
    mine = shard_seeds(train, rank, world).to(dev)
    nb = mine.numel() // B
    d = eng.depth
    for b in range(min(d, nb)):
        eng.prefetch(mine[b * B:(b + 1) * B], batch_counter(0, nb, b, rank, world))
    losses = []
    for b in range(nb):
        nxt = mine[(b + d) * B:(b + d + 1) * B] if b + d < nb else None
        loss = eng.pipelined_step(nxt, batch_counter(0, nb, b + d, rank, world) if nxt is not None else None)
        losses.append(loss.clone() if long_run else float(loss))
    if not long_run:
        assert int(eng.step_dev) == nb - 1        # the last Adam is deferred until flush()
    eng.flush()
    torch.cuda.synchronize()
    out[rank] = (eng.flat.cpu(), [float(x) for x in losses], int(eng.step_dev))
    dist.destroy_process_group()


def test_engine_data_parallel_many_steps_staged_allreduce(dev):
    """80 pipelined steps on two ranks without reading anything back in between.  With gloo on
    a device tensor the all-reduce is staged through pinned host memory and synchronises the
    stream every step (graphsage_amd/dist.py), so this covers the engine's deferred-Adam control
    flow and the replicas' bit-identity over a long run, NOT the no-host-sync regime RCCL runs in
    (that one has no single-GPU rehearsal; the async work-handle path is driven on CPU tensors
    in tests/test_dist_cpu.py)."""
    world = 2
    out = mp.Manager().dict()
    port = 29300 + (os.getpid() % 300)
    mp.spawn(_worker, args=(world, port, out, True), nprocs=world, join=True)
    flat0, loss0, steps0 = out[0]
    flat1, loss1, steps1 = out[1]
    assert steps0 == steps1 == LONG_STEPS
    assert torch.equal(flat0, flat1)
    assert all(x == x and 0.0 < x < 10.0 for x in loss0 + loss1)
    assert sum(loss0[-8:]) < sum(loss0[:8])                # and it trains


@pytest.mark.parametrize("aggregator", ["mean", "max_pool", "lstm"])
def test_engine_data_parallel_two_ranks(dev, aggregator):
    """(max_pool: the layer-0 "gather" starts with the pool Linear, which reads the weights — the
    deferred Adam step of the previous batch has to be applied in front of it, not behind it.)"""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29600 + (os.getpid() % 300)
    mp.spawn(_worker, args=(world, port, out, False, aggregator), nprocs=world, join=True)
    flat0, loss0, steps0 = out[0]
    flat1, loss1, steps1 = out[1]
    assert steps0 == steps1 == STEPS
    assert torch.equal(flat0, flat1)                       # replicas bit-identical
    assert loss0 != loss1                                  # ... on different batches
    # single-process emulation: both ranks' gradients, averaged, one Adam step
    from graphsage_amd import ops
    from graphsage_amd.dist import batch_counter, shard_seeds
    data, model, train = _setup(aggregator)
    eng = _engine(data, model, dev, keep_grads=True)
    shards = [shard_seeds(train, r, world).to(dev) for r in range(world)]
    for b in range(STEPS):
        total = torch.zeros_like(eng.grad)
        for r in range(world):
            eng.sample(shards[r][b * B:(b + 1) * B], batch_counter(0, STEPS, b, r, world), slot=0)
            eng.forward_backward(slot=0)
            total += eng.grad
            want = (loss0, loss1)[r][b]
            assert abs(float(eng.loss) - want) <= 1e-5 * max(1.0, abs(want))
        eng.grad.copy_(total)
        ops.adam_step(eng.flat, eng.grad, eng.exp_avg, eng.exp_avg_sq, eng._steps[:1], 1e-2,
                      grad_scale=1.0 / world)
    if aggregator == "lstm":      # cell weights with gradients near Adam's eps move by up to lr on a last-bit difference
        off = (eng.flat.cpu() - flat0).abs() > 1e-6 + 1e-4 * flat0.abs()
        assert int(off.sum()) <= off.numel() // 200, int(off.sum())      # (the differences compound over the steps)
    else:
        assert_trajectories_close(eng.flat.cpu(), flat0)


def _rccl_one_rank_worker(rank, world, port, out, aggregator):
    """One rank, backend nccl (= RCCL): the engine's collective branch — asynchronous all-reduce
    on RCCL's stream, work.wait() on the main stream, deferred Adam, the host's run-ahead bound —
    on the hardware backend, with no host synchronisation between the steps."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from graphsage_amd.dist import batch_counter
    data, model, _ = _setup(aggregator)
    seeds = torch.randint(0, 1500, (B * LONG_STEPS,), generator=torch.Generator().manual_seed(2)).to(dev)
    import copy
    res = {}
    for name, kw in (("plain", {}), ("collective", dict(force_collective=True, process_group=dist.group.WORLD))):
        eng = _engine(data, copy.deepcopy(model), dev, **kw)
        eng.trace = []
        d = eng.depth
        for b in range(d):
            eng.prefetch(seeds[b * B:(b + 1) * B], batch_counter(0, LONG_STEPS, b, 0, 1))
        losses = []
        for b in range(LONG_STEPS):
            nxt = seeds[(b + d) * B:(b + d + 1) * B] if b + d < LONG_STEPS else None
            loss = eng.pipelined_step(nxt, batch_counter(0, LONG_STEPS, b + d, 0, 1) if nxt is not None else None)
            losses.append(loss.clone())
        if name == "collective":
            assert eng._pending.active           # the last Adam is deferred until flush()
        eng.flush()
        torch.cuda.synchronize()
        res[name] = (eng.flat.cpu(), [float(x) for x in losses], int(eng.step_dev), list(eng.trace),
                     eng.overlap_gather)
    out[0] = res
    dist.destroy_process_group()


@pytest.mark.parametrize("aggregator", ["mean", "max_pool"])
def test_engine_collective_branch_on_rccl_one_rank(dev, aggregator):
    out = mp.Manager().dict()
    port = 28300 + (os.getpid() % 300)
    mp.spawn(_rccl_one_rank_worker, args=(1, port, out, aggregator), nprocs=1, join=True)
    plain, coll = out[0]["plain"], out[0]["collective"]
    assert plain[2] == coll[2] == LONG_STEPS
    assert not coll[4]                                     # the collective selects the serial schedule
    # same batches, same arithmetic (float atomics: last bits differ between any two runs)
    torch.testing.assert_close(torch.tensor(coll[1]), torch.tensor(plain[1]), rtol=2e-4, atol=1e-5)
    torch.testing.assert_close(coll[0], plain[0], rtol=2e-3, atol=2e-5)
    assert sum(coll[1][-8:]) < sum(coll[1][:8])
    # order in which a steady-state step is enqueued: the all-reduce is launched behind the
    # backward, the next batch's weight-free layer-0 gather goes between that launch and the
    # wait, Adam behind the wait, the forward contraction behind Adam.  (With a pooling
    # aggregator the layer-0 "gather" reads the pool weights: wait and Adam come first.)
    tr = coll[3]
    first = tr.index("allreduce_launch")
    steady = tr[first:]
    per_step = (["allreduce_launch", "gather0", "allreduce_wait", "adam", "forward_backward"] if aggregator == "mean"
                else ["allreduce_launch", "allreduce_wait", "adam", "gather0", "forward_backward"])
    n = (len(steady) - 1) // len(per_step)
    assert n >= LONG_STEPS - 2
    assert steady[:n * len(per_step)] == per_step * n


def _trainer_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch.nn.functional as F
    from graphsage_amd.datasets import planted_partition
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import UniformLoader
    from graphsage_amd.trainers import SupervisedGraphSageTrainerForNodeLevelTask, dataloader_kwargs
    dev = torch.device("cuda:0")
    data = planted_partition(1500, 30000, 24, 5, seed=8)
    n = data.num_nodes
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(0))
    for name, sl in (("train_mask", perm[:600]), ("val_mask", perm[600:900]), ("test_mask", perm[900:])):
        m = torch.zeros(n, dtype=torch.bool)
        m[sl] = True
        setattr(data, name, m)
    torch.manual_seed(3)
    model = GraphSAGE(24, 16, "max_pool", 2, 5).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    dataloader_kwargs['batch_size'] = 100
    tr = SupervisedGraphSageTrainerForNodeLevelTask(data, UniformLoader, supervised_only=True, use_engine=False,
                                                    loss_fn=F.cross_entropy, model=model, optimizer=opt,
                                                    num_epochs=2, device=dev, k1=5, k2=3)
    best = tr.run(verbose=False)
    seeds = tr.train_loader._idx_dev.cpu().tolist()
    out[rank] = (torch.cat([p.detach().reshape(-1).cpu() for p in model.parameters()]), seeds, best['test_f1'])
    dist.destroy_process_group()


def test_module_path_trainer_data_parallel_two_ranks(dev):
    """Any aggregator under data parallelism goes through the module path: seeds sharded over the
    ranks, gradients averaged with one flat all-reduce (graphsage_amd.dist.all_reduce_gradients)
    — max_pool model, two ranks on the one GPU: replicas stay bit-identical and learn."""
    world = 2
    out = mp.Manager().dict()
    port = 29100 + (os.getpid() % 150)
    mp.spawn(_trainer_worker, args=(world, port, out), nprocs=world, join=True)
    w0, s0, f0 = out[0]
    w1, s1, f1 = out[1]
    assert torch.equal(w0, w1)
    assert len(s0) == len(s1) == 300 and not set(s0) & set(s1)
    assert f0 == f1 and f0 > 0.5


def _unsup_trainer_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from graphsage_amd.datasets import planted_partition
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import UniformLoader
    from graphsage_amd.trainers import UnsupervisedGraphSageTrainerForNodeLevelTask, dataloader_kwargs
    dev = torch.device("cuda:0")
    data = planted_partition(1500, 30000, 24, 5, seed=8, noise=1.0)
    n = data.num_nodes
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(0))
    for name, sl in (("train_mask", perm[:600]), ("val_mask", perm[600:900]), ("test_mask", perm[900:])):
        m = torch.zeros(n, dtype=torch.bool)
        m[sl] = True
        setattr(data, name, m)
    torch.manual_seed(3)
    model = GraphSAGE(24, 16, "mean", 2).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=5e-3)
    dataloader_kwargs['batch_size'] = 100
    tr = UnsupervisedGraphSageTrainerForNodeLevelTask(data, UniformLoader, model=model, optimizer=opt,
                                                      num_epochs=2, device=dev, k1=5, k2=3)
    best = tr.run(verbose=False)
    out[rank] = (torch.cat([p.detach().reshape(-1).cpu() for p in model.parameters()]),
                 len(tr.train_loader.dataset), len(tr.train_eval_loader.dataset), best)
    dist.destroy_process_group()


def test_unsupervised_trainer_data_parallel_two_ranks(dev):
    """run() of the unsupervised trainer under data parallelism: training seeds are sharded, the
    downstream head is fitted on the embeddings of the WHOLE training split (its labels are the
    whole split's), replicas stay bit-identical and report the same metrics."""
    world = 2
    out = mp.Manager().dict()
    port = 28700 + (os.getpid() % 200)
    mp.spawn(_unsup_trainer_worker, args=(world, port, out), nprocs=world, join=True)
    (w0, nt0, ne0, b0), (w1, nt1, ne1, b1) = out[0], out[1]
    assert torch.equal(w0, w1)
    assert nt0 == nt1 == 300 and ne0 == ne1 == 600
    assert b0['test_f1'] == b1['test_f1'] and b0['test_f1'] > 0.5 and 'train_f1' in b0
