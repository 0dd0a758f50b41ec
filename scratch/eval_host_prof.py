"""Host-side profile of engine.evaluate() over Reddit's 55,703 test nodes (109 batches): where the ~146 us per batch go."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphsage_amd.datasets import reddit_like_device
from graphsage_amd.engine import SageTrainEngine
from graphsage_amd.models import GraphSAGE
from graphsage_amd.samplers import NeighborSampler
from graphsage_amd.samplers.uniform_loader import device_graph

dev = torch.device("cuda", 0)
data = reddit_like_device(232965, 114615892, 602, 41, seed=0, device=dev)
graph = device_graph(data)
torch.manual_seed(0)
agg = sys.argv[1] if len(sys.argv) > 1 else "mean"
model = GraphSAGE(602, 256, agg, 2, 41).to(dev)
eng = SageTrainEngine(model, NeighborSampler.from_graph(graph, [25, 10], seed=0), graph.features(data),
                      graph.labels(data), 512, lr=1e-3, overlap_gather='auto')
ids = torch.randperm(232965, generator=torch.Generator().manual_seed(1))[:55703].to(dev)
for _ in range(2):
    eng.evaluate(ids)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.evaluate(ids)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("evaluate: host %.2f ms, complete %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
eng.evaluate(ids)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
