"""Debug: the engine's deferred all-reduce path on the REAL backend (nccl = RCCL) with the one
rank a one-GPU box allows: world_size is forced to 2 inside the engine so that every step issues
dist.all_reduce(async_op=True) on RCCL's stream and applies Adam behind work.wait()."""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(150, exit=True)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

import graphsage_amd.engine as engine_mod  # noqa: E402
from graphsage_amd.datasets import planted_partition  # noqa: E402
from graphsage_amd.engine import SageTrainEngine  # noqa: E402
from graphsage_amd.models import GraphSAGE  # noqa: E402
from graphsage_amd.samplers import NeighborSampler  # noqa: E402
from graphsage_amd.samplers.uniform_loader import device_graph  # noqa: E402

calls = {"n": 0}


def forced(flat, group=None, async_op=False):
    calls["n"] += 1
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work if async_op else flat


engine_mod.all_reduce_flat = forced
B, FAN, steps = 256, [10, 5], int(sys.argv[1]) if len(sys.argv) > 1 else 300
data = planted_partition(20000, 400000, 64, 7, seed=8)
torch.manual_seed(3)
model = GraphSAGE(64, 128, "mean", 2, 7)
graph = device_graph(data)
eng = SageTrainEngine(model.to(dev), NeighborSampler(data, FAN, seed=5), graph.features(data),
                      graph.labels(data), B, lr=1e-2, world_size=1, process_group=dist.group.WORLD)
eng.world_size = 2          # take the data-parallel branches (grad_scale becomes 1/2)
seeds = torch.randint(0, 20000, (B * (steps + 8),), generator=torch.Generator().manual_seed(1)).to(dev)
d = eng.depth
for b in range(d):
    eng.prefetch(seeds[b * B:(b + 1) * B], b)
losses = []
t0 = time.perf_counter()
for b in range(steps):
    losses.append(eng.pipelined_step(seeds[(b + d) * B:(b + d + 1) * B], b + d).clone())
eng.flush()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ls = [float(x) for x in losses]
print(f"{steps} steps in {dt:.3f}s ({dt / steps * 1e3:.3f} ms/step), {calls['n']} all-reduces on nccl, "
      f"loss {ls[0]:.4f} -> {ls[-1]:.4f}, adam steps {int(eng.step_dev)}", flush=True)
assert calls["n"] == steps and int(eng.step_dev) == steps and ls[-1] < ls[0] and all(x == x for x in ls)
dist.destroy_process_group()
print("OK")
