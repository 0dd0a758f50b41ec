"""Host-side profile of the products-scale unsupervised step (serial schedule, forked variants)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphsage_amd.datasets import reddit_like_device
from graphsage_amd.engine import SageTrainEngine
from graphsage_amd.models import GraphSAGE
from graphsage_amd.samplers import NeighborSampler
from graphsage_amd.samplers.uniform_loader import device_graph

dev = torch.device("cuda", 0)
N, E, F, H, B = 2400000, 62000000, 100, 256, 512
data = reddit_like_device(N, E, F, 47, seed=0, device=dev)
graph = device_graph(data)
torch.manual_seed(0)
model = GraphSAGE(F, H, "mean", 2, 47).to(dev)
eng = SageTrainEngine(model, NeighborSampler.from_graph(graph, [25, 10], seed=0), graph.features(data),
                      graph.labels(data), B, lr=1e-3, overlap_gather='auto', unsup=False)
perm = torch.randperm(N, generator=torch.Generator().manual_seed(1)).to(dev)
seeds = lambda s: perm[(s * B) % (N - B):(s * B) % (N - B) + B]
for d in range(eng.depth):
    eng.prefetch(seeds(d), d)
def run(n, s0):
    for s in range(s0, s0 + n):
        eng.pipelined_step(seeds(s + eng.depth), s + eng.depth)
run(300, 0)
torch.cuda.synchronize()
t0 = time.perf_counter(); run(1000, 300); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host %.1f us per step, complete %.1f us per step" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); run(1000, 1300); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
