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
