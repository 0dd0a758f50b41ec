"""Does the step time depend on WHERE the engine's buffers sit?  Engines built one after the other in one
process (torch's caching allocator hands out different addresses), 600 pipelined steps each, with the addresses
of the large buffers."""
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
N, E, F, C, n_train, B = 232965, 114615892, 602, 41, 153431, 512
data = reddit_like_device(N, E, F, C, seed=0, device=dev)
graph = device_graph(data)
feats, labels = graph.features(data), graph.labels(data)
data.edge_index = None
train = torch.randperm(N, generator=torch.Generator().manual_seed(0))[:n_train].to(dev)
nb = n_train // B
seeds = lambda s: train[(s % nb) * B:(s % nb + 1) * B]
keep = []
for it in range(int(os.environ.get("SAGE_PLACEMENT_ITERS", "12"))):
    if it % 3 == 2:
        keep.append(torch.empty((it + 1) * 3 * 1024 * 1024 + 4096 * it, dtype=torch.uint8, device=dev))   # shifts what follows
    torch.manual_seed(0)
    model = GraphSAGE(F, 256, agg, 2, C).to(dev)
    eng = SageTrainEngine(model, NeighborSampler.from_graph(graph, [25, 10], seed=0), feats, labels, B, lr=1e-3,
                          overlap_gather=ov)
    d = eng.depth
    for s in range(d):
        eng.prefetch(seeds(s), s)
    for s in range(300):
        eng.pipelined_step(seeds(s + d), s + d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(300, 900):
        eng.pipelined_step(seeds(s + d), s + d)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 600 * 1e3
    while eng.primed:
        eng.pipelined_step()
    eng.flush()
    torch.cuda.synchronize()
    ptrs = {"side": eng.side.cuda_stream, "samp": eng.samp.cuda_stream}
    print("%.4f ms " % ms + " ".join("%s=%x" % (k, v & 0xffffffffff) for k, v in ptrs.items()), flush=True)
