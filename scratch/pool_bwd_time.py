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
dagg = ops.empty_rows(a['dagg'].size(0), a['dagg'].size(1), dev)
dagg.copy_(torch.randn(a['dagg'].shape, device=dev))
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
    print("max", name, "form", os.environ.get("SAGE_POOL_BWD", "2"), "us:", " ".join("%.0f" % t for t in ts[2:]),
          "checksum %.6e" % float(a['dpre'].double().abs().sum()))
P = a['P']
for name, ws in (("atomic", None), ("indeg", indeg)):
    ts = []
    for it in range(12):
        a['dpre'].zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.pool_mean_bwd(dagg, P, buf.seg_ptr, buf.src, t_cap, a['dpre'], num_targets_dev=t_dev, indeg=ws)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print("mean", name, "form", os.environ.get("SAGE_POOL_BWD", "2"), "us:", " ".join("%.0f" % t for t in ts[2:]),
          "checksum %.6e" % float(a['dpre'].double().abs().sum()))

# ---- pulled by the source rows (sage_pool_reverse_edges + sage_pool_*_bwd_pull)
def timed(fn, reps=12):
    ts = []
    for it in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return " ".join("%.0f" % t for t in ts[2:])

s_cap = a['dpre'].size(0)
s_dev = eng.n_dev[0][a['hop'] + 1] if a['hop'] + 1 < len(eng.n_dev[0]) else None
print("s_cap", s_cap, "n_dev", [int(v) for v in eng.n_dev[0]])
rev = ops.ReverseEdges(s_cap, buf.src.numel(), dev)
print("reverse lists us:", timed(lambda: rev.build(buf.seg_ptr, buf.src, t_cap, num_targets_dev=t_dev)))
F = a['P'].size(1)
arg8 = torch.zeros(t_cap, (F + 3) // 4 * 4, dtype=torch.uint8, device=dev)
agg8 = torch.zeros_like(eng.agg0[0])
print("max fwd int64 us:", timed(lambda: ops.gather_reduce_fwd(P, buf.seg_ptr, buf.src, t_cap, 'max', num_targets_dev=t_dev,
                                                                 out=agg8, argmax=a['argmax'])))
print("max fwd arg8 us:", timed(lambda: ops.pool_max_fwd(P, buf.seg_ptr, buf.src, t_cap, 25, agg8, arg8, num_targets_dev=t_dev)))
a['dpre'].zero_()
ops.pool_max_bwd(dagg, agg8, buf.seg_ptr, buf.src, a['argmax'], t_cap, a['dpre'], num_targets_dev=t_dev)
want = a['dpre'].clone()
nsrc = torch.tensor([st[0]], dtype=torch.int64, device=dev)
print("max pull us:", timed(lambda: ops.pool_max_bwd_pull(dagg, arg8, rev, a['dpre'], num_sources_dev=nsrc)),
      "max diff %.3e" % float((a['dpre'][:st[0]] - want[:st[0]]).abs().max()))
a['dpre'].zero_()
ops.pool_mean_bwd(dagg, P, buf.seg_ptr, buf.src, t_cap, a['dpre'], num_targets_dev=t_dev)
want = a['dpre'].clone()
print("mean pull us:", timed(lambda: ops.pool_mean_bwd_pull(dagg, P, buf.seg_ptr, rev, a['dpre'], num_sources_dev=nsrc)),
      "max diff %.3e" % float((a['dpre'][:st[0]] - want[:st[0]]).abs().max()))
