"""Times sage_pool_max_bwd on one real Reddit-shaped max_pool batch (layer 0: ~13.1k targets,
~81k source rows, F = 602): all-atomic route vs in-degree route; dpre zeroed outside the events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphsage_amd import ops
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
seeds = torch.randperm(232965, generator=torch.Generator().manual_seed(2))[:512].to(dev)
eng.step(seeds, counter=5)
torch.cuda.synchronize()
a, buf = eng.act[0], eng.buf
t_cap, t_dev = a['t_cap'], eng.n_dev[0][a['hop']]
st = buf.state.tolist()
print("nodes", st[0], "edges", st[1], "targets", int(t_dev))
indeg = a['indeg']
print("rows with 1 edge:", int((indeg == 1).sum()), " >1:", int((indeg > 1).sum()),
      " edges into shared rows:", int(indeg[indeg > 1].sum()))
dagg = torch.randn_like(a['dagg'])
db = torch.zeros(608, device=dev)
for name, ws in (("atomic", None), ("indeg", indeg)):
    ts = []
    for it in range(12):
        a['dpre'].zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.pool_max_bwd(dagg, eng.agg0[0], buf.seg_ptr, buf.src, a['argmax'], t_cap, a['dpre'], dbias=None,
                         num_targets_dev=t_dev, indeg=ws)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(name, os.environ.get("SAGE_POOL_BWD_VEC", ""), "us:", " ".join("%.0f" % t for t in ts[2:]))
