import copy, sys, torch
sys.path.insert(0, '.')
from graphsage_amd.datasets import planted_partition
from graphsage_amd.engine import SageTrainEngine
from graphsage_amd.models import GraphSAGE
from graphsage_amd.samplers import NeighborSampler
from graphsage_amd.samplers.uniform_loader import device_graph
dev = torch.device('cuda:0')
data = planted_partition(2000, 40000, 24, 5, seed=4)
B = 32
graph = device_graph(data)
torch.manual_seed(1)
model_a = GraphSAGE(24, 16, "mean", 2, 5).to(dev)
model_b = copy.deepcopy(model_a)
mk = lambda m: SageTrainEngine(m, NeighborSampler(data, [4, 3], seed=9), graph.features(data), graph.labels(data), B, lr=1e-2)
eager, eng = mk(model_a), mk(model_b)
pool = torch.randperm(2000, generator=torch.Generator().manual_seed(0))[:3 * B + 7].to(dev)
eng.step(pool[:B], counter=1000)
eager.step(pool[:B], counter=1000)
eng.capture(pool, counter0=5, counter_stride=2, split_gather=False)
for s in range(7):
    b = s % 3
    la = eager.step(pool[b * B:(b + 1) * B], counter=5 + 2 * s).clone()
    lb = eng.replay().clone()
    st_e = eager.buf.state.tolist(); st_g = eng.buf.state.tolist()
    print(s, 'eager', [hex(v) for v in st_e[:6]], 'graph', [hex(v) for v in st_g[:6]], 'other', [hex(v) for v in eng.bufs[1 - eng.last_slot].state.tolist()[:6]], float(la), float(lb))
