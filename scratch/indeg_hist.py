"""In-degree distribution of the layer-0 source rows of Reddit-shaped batches (how long the reverse lists get)."""
import os, sys
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
model = GraphSAGE(602, 256, "max_pool", 2, 41).to(dev)
eng = SageTrainEngine(model, NeighborSampler.from_graph(graph, [25, 10], seed=0), graph.features(data),
                      graph.labels(data), 512, lr=1e-3, keep_grads=True)
for k in range(3):
    seeds = torch.randperm(232965, generator=torch.Generator().manual_seed(2 + k))[:512].to(dev)
    eng.step(seeds, counter=5 + k)
    torch.cuda.synchronize()
    a, buf = eng.act[0], eng.buf
    st = buf.state.tolist()
    indeg = a['indeg'][:st[0]].long()
    q = torch.tensor([0.5, 0.9, 0.99, 0.999])
    print("nodes", st[0], "edges", st[1], "targets", int(eng.n_dev[0][a['hop']]), "indeg max", int(indeg.max()),
          "quantiles", torch.quantile(indeg.float().cpu(), q).tolist(), "zero rows", int((indeg == 0).sum()),
          "rows >32:", int((indeg > 32).sum()), "edges in them", int(indeg[indeg > 32].sum()),
          "first row >32 at", int((indeg > 32).nonzero()[0]) if (indeg > 32).any() else -1)
    deg = (buf.seg_ptr[1:int(eng.n_dev[0][a['hop']]) + 1] - buf.seg_ptr[:int(eng.n_dev[0][a['hop']])])
    print("  target degree: max", int(deg.max()), "mean %.1f" % float(deg.float().mean()))
