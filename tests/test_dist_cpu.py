"""N > 1 host logic on CPU: gloo, world_size 2 — seed sharding, per-rank Philox counters, and
the flat-gradient all-reduce + identical Adam step that keeps replicas bit-identical."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sage_ref


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from graphsage_amd.dist import all_reduce_flat, batch_counter, shard_seeds
    idx = torch.randperm(1001, generator=torch.Generator().manual_seed(0))
    mine = shard_seeds(idx, rank, world)
    counters = [batch_counter(e, 7, b, rank, world) for e in range(2) for b in range(7)]
    # each rank's "gradient" differs; after the all-reduce + Adam(grad/world) params must agree
    g = torch.Generator().manual_seed(100 + rank)
    p = torch.ones(50)
    m, v = torch.zeros(50), torch.zeros(50)
    for step in range(1, 4):
        grad = torch.randn(50, generator=g)
        all_reduce_flat(grad)
        p, m, v = sage_ref.adam_ref(p, grad / world, m, v, step, 1e-2)
    gathered = [torch.zeros(50) for _ in range(world)]
    dist.all_gather(gathered, p)
    # module-path helper: gradients averaged over ranks (one parameter has no gradient on rank 1)
    from graphsage_amd.dist import all_reduce_gradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    x = torch.full((5, 4), float(rank + 1))
    h = net[0](x)
    (net[1](h).sum() if rank == 0 else h.sum()).backward()
    local = [None if q.grad is None else q.grad.clone() for q in net.parameters()]
    all_reduce_gradients(net)
    grads = [q.grad.clone() for q in net.parameters()]
    out[rank] = (mine.tolist(), counters, [t.tolist() for t in gathered], local, grads)
    dist.destroy_process_group()


def test_gloo_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert len(a[0]) == len(b[0]) == 500 and not set(a[0]) & set(b[0])     # equal, disjoint shards
    assert not set(a[1]) & set(b[1]) and len(set(a[1])) == 14                 # distinct Philox streams
    assert a[2][0] == a[2][1] == b[2][0] == b[2][1]                           # replicas bit-identical
    for la, lb, ga, gb in zip(a[3], b[3], a[4], b[4]):
        lb = torch.zeros_like(la) if lb is None else lb
        assert torch.equal(ga, gb)
        torch.testing.assert_close(ga, (la + lb) / 2)


def _pending_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from graphsage_amd.dist import PendingReduce
    g = torch.Generator().manual_seed(7 + rank)
    p = torch.ones(64)
    m, v = torch.zeros(64), torch.zeros(64)
    pend = PendingReduce()
    grad = torch.zeros(64)
    applied = 0
    handles = []
    for step in range(1, 13):
        # the engine's order (pipelined_step): work that needs no weights, THEN the deferred
        # optimizer step of the previous batch, then this batch's gradient and its all-reduce
        weight_free = torch.randn(64, generator=g).sum()          # stands for sampler wait + gather
        if pend.complete():
            applied += 1
            p, m, v = sage_ref.adam_ref(p, grad / world, m, v, applied, 1e-2)
        grad = torch.randn(64, generator=g) + 0.1 * p + 0.0 * weight_free
        pend.launch(grad, None)
        handles.append(pend.work is not None and hasattr(pend.work, "wait"))
        try:
            pend.launch(grad, None)                               # a second launch must be refused
            raise AssertionError("double launch accepted")
        except RuntimeError:
            pass
    assert pend.complete() and not pend.complete()                # flush(): exactly one pending
    applied += 1
    p, m, v = sage_ref.adam_ref(p, grad / world, m, v, applied, 1e-2)
    out[rank] = (p.tolist(), applied, all(handles))
    dist.destroy_process_group()


def test_pending_reduce_async_handle_across_steps():
    """The engine's data-parallel control flow with a REAL asynchronous work handle (gloo on CPU
    tensors, async_op=True): the all-reduce launched at the end of step i is completed — and its
    Adam step applied — inside step i+1; replicas stay bit-identical and equal a serial emulation."""
    world = 2
    out = mp.Manager().dict()
    port = 29000 + (os.getpid() % 400)
    mp.spawn(_pending_worker, args=(world, port, out), nprocs=world, join=True)
    (p0, n0, h0), (p1, n1, h1) = out[0], out[1]
    assert n0 == n1 == 12 and h0 and h1
    assert p0 == p1
    # serial emulation of both ranks
    gens = [torch.Generator().manual_seed(7 + r) for r in range(world)]
    p = torch.ones(64)
    m, v = torch.zeros(64), torch.zeros(64)
    total = None
    for step in range(1, 13):
        wf = [torch.randn(64, generator=g).sum() for g in gens]
        if total is not None:
            p, m, v = sage_ref.adam_ref(p, total / world, m, v, step - 1, 1e-2)
        total = sum(torch.randn(64, generator=g) + 0.1 * p + 0.0 * w for g, w in zip(gens, wf))
    p, m, v = sage_ref.adam_ref(p, total / world, m, v, 12, 1e-2)
    torch.testing.assert_close(torch.tensor(p0), p, rtol=1e-6, atol=1e-7)
