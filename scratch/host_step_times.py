"""Host time of each pipelined_step() of the headline schedule: percentiles over 3000 steps.  If the
host is held back by the launch queue the distribution is bimodal (fast natural steps, long blocked ones)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd.datasets import reddit_like_device
from graphsage_amd.engine import SageTrainEngine
from graphsage_amd.models import GraphSAGE
from graphsage_amd.samplers import NeighborSampler
from graphsage_amd.samplers.uniform_loader import device_graph

dev = torch.device("cuda", 0)
N, E, F, C, n_train, B = 232965, 114615892, 602, 41, 153431, 512
data = reddit_like_device(N, E, F, C, seed=0, device=dev)
graph = device_graph(data)
feats, labels = graph.features(data), graph.labels(data)
torch.manual_seed(0)
model = GraphSAGE(F, 256, "mean", 2, C).to(dev)
data.edge_index = None
eng = SageTrainEngine(model, NeighborSampler.from_graph(graph, [25, 10], seed=0), feats, labels, B, lr=1e-3,
                      overlap_gather=True, group=4)
train = torch.randperm(N, generator=torch.Generator().manual_seed(0))[:n_train].to(dev)
nb = train.numel() // B
seeds = lambda s: train[(s % nb) * B:(s % nb + 1) * B]
d = eng.depth
for s in range(d):
    eng.prefetch(seeds(s), s)
for s in range(600):
    eng.pipelined_step(seeds(s + d), s + d)
torch.cuda.synchronize()
ts = []
t00 = time.perf_counter()
for s in range(600, 3600):
    t0 = time.perf_counter()
    eng.pipelined_step(seeds(s + d), s + d)
    ts.append(time.perf_counter() - t0)
t_issue = time.perf_counter() - t00
torch.cuda.synchronize()
t_all = time.perf_counter() - t00
ts = torch.tensor(ts) * 1e6
q = torch.quantile(ts, torch.tensor([0.05, 0.25, 0.5, 0.75, 0.9, 0.99]))
print("per-step host us: 5/25/50/75/90/99 % =", [round(float(x), 1) for x in q], "mean", round(float(ts.mean()), 1))
print("issue %.4f ms/step, complete %.4f ms/step" % (t_issue / 3, t_all / 3))
# the same calls with the GPU idle in between (no queue pressure): natural host cost
ts2 = []
for s in range(3600, 3900):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.pipelined_step(seeds(s + d), s + d)
    ts2.append(time.perf_counter() - t0)
ts2 = torch.tensor(ts2) * 1e6
print("with a synchronize before each step: median %.1f us, mean %.1f us" % (float(ts2.median()), float(ts2.mean())))
