"""One evaluation pass (engine.evaluate over Reddit's 55,703 test nodes, 109 batches of 512) after a short
training run — for a kernel trace of the forward-only pipeline (scratch/prof_r3.sh) and its wall time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphsage_amd.datasets import reddit_like_device
from graphsage_amd.engine import SageTrainEngine
from graphsage_amd.models import GraphSAGE
from graphsage_amd.samplers import NeighborSampler
from graphsage_amd.samplers.uniform_loader import device_graph

agg = sys.argv[1] if len(sys.argv) > 1 else "mean"
ov = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
dev = torch.device("cuda", 0)
N, E, F, C, B = 232965, 114615892, 602, 41, 512
data = reddit_like_device(N, E, F, C, seed=0, device=dev)
graph = device_graph(data)
feats, labels = graph.features(data), graph.labels(data)
data.edge_index = None
perm = torch.randperm(N, generator=torch.Generator().manual_seed(0)).to(dev)
train, test = perm[:153431], perm[-55703:]
torch.manual_seed(0)
model = GraphSAGE(F, 256, agg, 2, C).to(dev)
eng = SageTrainEngine(model, NeighborSampler.from_graph(graph, [25, 10], seed=0), feats, labels, B, lr=1e-3,
                      overlap_gather=ov)
eng.train_epoch(train[:100 * B], epoch=0)
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    correct, count = eng.evaluate(test, counter0=1 << 20)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{agg} overlap={int(ov)} pass {rep}: {dt*1e3:.2f} ms for {int(count)} nodes = {dt/109*1e3:.4f} ms per batch, "
          f"accuracy {int(correct)/int(count):.4f}", flush=True)
