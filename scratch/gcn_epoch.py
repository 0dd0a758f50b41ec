"""train_epoch() timing for one aggregator, overlapped against one-stream schedule, with and without an
evaluation pass between the epochs (the fig2a harness's loop without the harness)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd.datasets import reddit_like_device
from graphsage_amd.engine import SageTrainEngine
from graphsage_amd.models import GraphSAGE
from graphsage_amd.samplers import NeighborSampler
from graphsage_amd.samplers.uniform_loader import device_graph

agg = sys.argv[1] if len(sys.argv) > 1 else "gcn"
dev = torch.device("cuda", 0)
N, E, F, C, n_train, B = 232965, 114615892, 602, 41, 153431, 512
data = reddit_like_device(N, E, F, C, seed=0, device=dev)
graph = device_graph(data)
feats, labels = graph.features(data), graph.labels(data)
data.edge_index = None
perm = torch.randperm(N, generator=torch.Generator().manual_seed(0)).to(dev)
train, test = perm[:n_train], perm[n_train:n_train + 55000]
for ov in (True, False, True, False):
    for ev in (False, True):
        torch.manual_seed(0)
        model = GraphSAGE(F, 256, agg, 2, C).to(dev)
        eng = SageTrainEngine(model, NeighborSampler.from_graph(graph, [25, 10], seed=0), feats, labels, B, lr=1e-3,
                              overlap_gather=ov)
        ts = []
        for ep in range(4):
            seeds = train[torch.randperm(n_train, device=dev)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.train_epoch(seeds, epoch=ep, drop_last=False)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / (n_train // B + 1) * 1e3)
            if ev:
                eng.evaluate(test)
        print(agg, "overlap" if ov else "serial ", "eval" if ev else "    ", " ".join("%.4f" % t for t in ts), flush=True)
