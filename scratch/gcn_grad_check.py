"""Diagnostic: gcn engine step at Reddit size — engine grads and module-path grads against a float64
autograd reference built from plain torch ops."""
import copy, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from graphsage_amd.datasets import reddit_like_device
from graphsage_amd.samplers.uniform_loader import device_graph
from graphsage_amd.engine import SageTrainEngine
from graphsage_amd.models import GraphSAGE
from graphsage_amd.samplers import NeighborSampler
N, E, FEAT, C, B, FAN = 232965, 114615892, 602, 41, 512, [25, 10]
dev = torch.device('cuda:0')
data = reddit_like_device(N, E, FEAT, C, seed=0, device=dev)
graph = device_graph(data)
for agg_name in ("gcn",):
    torch.manual_seed(0)
    model = GraphSAGE(FEAT, 256, agg_name, 2, C).to(dev)
    ref_model = copy.deepcopy(model)
    sampler = NeighborSampler.from_graph(graph, FAN, seed=0)
    engine = SageTrainEngine(model, sampler, graph.features(data), graph.labels(data), B, lr=1e-3, keep_grads=True)
    seeds = torch.randperm(N, generator=torch.Generator().manual_seed(2))[:B].to(dev)
    before = [p.detach().clone() for p in model.parameters()]
    loss = engine.step(seeds, counter=11)
    buf = engine.buf
    st = buf.state.tolist()
    nn, ne = st[0], st[1]
    node, ei = buf.node[:nn], torch.stack([buf.src[:ne], buf.dst[:ne]])
    ref_model.prune_to_seeds = True
    y_hat = ref_model(data.x[node], ei)[:B]
    F.cross_entropy(y_hat, data.y[seeds]).backward()
    # float64 reference
    ws = [w.double().requires_grad_(True) for w in before]
    h = data.x[node].double()
    gcn = agg_name == "gcn"
    pairs = [(w, None) for w in ws] if gcn else list(zip(ws[0::2], ws[1::2]))
    zs, aggs = [], []
    for i, (wl, wr) in enumerate(pairs):
        agg = torch.zeros_like(h).index_add_(0, ei[1], h[ei[0]])
        deg = torch.bincount(ei[1], minlength=nn).unsqueeze(1)
        if gcn:
            z = F.linear(torch.where(deg > 0, (agg + h) / (deg + 1), torch.zeros_like(agg)), wl)
        else:
            z = F.linear(agg / deg.clamp(min=1), wl) + F.linear(h, wr)
        z.retain_grad()
        zs.append(z)
        aggs.append(torch.where(deg > 0, (agg + h) / (deg + 1), torch.zeros_like(agg)) if gcn else agg / deg.clamp(min=1))
        h = F.normalize(z, p=2.0, dim=-1)
        if i == 0:
            h = h.relu()
            h.retain_grad()
            h0 = h
    F.cross_entropy(h[:B], data.y[seeds]).backward()
    t1 = st[9]
    if gcn:
        from graphsage_amd import ops
        a1 = engine.act[1]
        dh = torch.zeros(engine.act[0]['t_cap'], 256, device=dev)
        ops.gather_reduce_bwd(a1['dagg'], buf.seg_ptr, buf.src, a1['t_cap'], 'gcn', dh)
        d_dh = (dh[:t1].double() - h0.grad[:t1]).abs().max(1).values
        bad = (d_dh > 1e-9).nonzero().flatten()
        print("recomputed dh0 from the engine's dagg1: max diff %.3g, bad rows %s" % (d_dh.max().item(), bad[:8].tolist()))
        colptr = graph.colptr
        for r in (10342, 11051):
            gid = int(node[r])
            print(" row", r, "global", gid, "in-degree", int(colptr[gid + 1] - colptr[gid]), "layer-0 seg len",
                  int(buf.seg_ptr[r + 1] - buf.seg_ptr[r]), "uses as layer-1 source", int((ei[0][:st[13] if len(st) > 13 else ne] == r).sum()),
                  "|dh ref| %.3g" % h0.grad[r].abs().max().item(), "|dz eng| %.3g |dz ref| %.3g" % (
                      engine.act[0]['dz'][r].abs().max().item(), zs[0].grad[r].abs().max().item()),
                  "y0 eng>0: %d ref>0: %d" % (int((engine.act[0]['y'][r] > 0).sum()), int((h0[r] > 0).sum())))
    a0 = engine.act[0]
    agg0 = engine._agg0(0, 0) if hasattr(engine, '_agg0') else a0['agg']
    d_agg = (agg0[:t1].double() - aggs[0][:t1]).abs().max(1).values
    d_dz = (a0['dz'][:t1].double() - zs[0].grad[:t1]).abs().max(1).values
    print(agg_name, "t1", t1, "agg0 max diff %.3g (rows > 1e-6: %d)  dz0 max diff %.3g (rows > 1e-9: %d, first %s) max|dz0| %.3g" % (
        d_agg.max().item(), int((d_agg > 1e-6).sum()), d_dz.max().item(), int((d_dz > 1e-9).sum()),
        (d_dz > 1e-9).nonzero().flatten()[:8].tolist(), zs[0].grad.abs().max().item()))
    for k, (p, q, w) in enumerate(zip(model.parameters(), ref_model.parameters(), ws)):
        g64 = w.grad
        ge, gm = engine.gviews[p].double(), q.grad.double()
        print(agg_name, k, tuple(p.shape), "max|g| %.3g  engine-ref %.3g  module-ref %.3g  engine-module %.3g" % (
            g64.abs().max().item(), (ge - g64).abs().max().item(), (gm - g64).abs().max().item(), (ge - gm).abs().max().item()))
