"""How much of the step is the main stream alone (no sampler beside it, no cross-stream waits)?"""
import sys, time, torch
sys.path.insert(0, '.')
from graphsage_amd.datasets import reddit_like_device
from graphsage_amd.engine import SageTrainEngine
from graphsage_amd.models import GraphSAGE
from graphsage_amd.samplers import NeighborSampler
from graphsage_amd.samplers.uniform_loader import device_graph
dev = torch.device('cuda:0')
data = reddit_like_device(232965, 114615892, 602, 41, seed=0, device=dev)
graph = device_graph(data)
feats, labels = graph.features(data), graph.labels(data)
data.edge_index = None
torch.manual_seed(0)
model = GraphSAGE(602, 256, 'mean', 2, 41).to(dev)
eng = SageTrainEngine(model, NeighborSampler.from_graph(graph, [25, 10], seed=0), feats, labels, 512)
seeds = torch.randperm(232965, device=dev)[:512 * 8]
eng.step(seeds[:512], 0)
torch.cuda.synchronize()
def timeit(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
def main_only():
    eng.forward_backward(0); eng.optimizer_step()
print(f"main stream only (gather..adam on one resident batch): {timeit(main_only):.1f} us")
i = [0]
def serial():
    i[0] += 1
    eng.step(seeds[(i[0] % 8) * 512:(i[0] % 8 + 1) * 512], i[0])
print(f"serial step (sampler + compute on one stream):        {timeit(serial):.1f} us")
eng.prefetch(seeds[:512], 1000)
def piped():
    i[0] += 1
    eng.pipelined_step(seeds[(i[0] % 8) * 512:(i[0] % 8 + 1) * 512], 1000 + i[0])
print(f"pipelined step (sampler on the side stream):          {timeit(piped):.1f} us")
# --- sync cost alone: pipelined step with the sampler calls removed (both slots hold valid batches)
while eng.queue:
    eng.pipelined_step()
torch.cuda.synchronize()
real_sample = eng.sample
eng.sample = lambda *a, **k: None
eng.prefetch(seeds[:512], 1)
print(f"pipelined, sampler kernels removed (events + waits only): {timeit(piped):.1f} us")
while eng.queue:
    eng.pipelined_step()
eng.sample = real_sample
