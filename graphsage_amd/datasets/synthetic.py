"""Seeded synthetic graphs with the shapes of the reference's datasets."""
import math

import torch

from ..data import Data

# name -> (num_nodes, num_directed_edges, num_features, num_classes, train, val, test)
SHAPES = {
    'Cora': (2708, 10556, 1433, 7, 140, 500, 1000),
    'CiteSeer': (3327, 9104, 3703, 6, 120, 500, 1000),
    'PubMed': (19717, 88648, 500, 3, 60, 500, 1000),
    'Reddit': (232965, 114615892, 602, 41, 153431, 23831, 55703),
    'Flickr': (89250, 899756, 500, 7, 44625, 22312, 22313),
}


def planted_partition(num_nodes, num_edges, num_features, num_classes, homophily=0.8,
                      noise=1.0, seed=0, device='cpu'):
    """Undirected planted-partition graph stored as a symmetric directed edge list
    (num_edges directed edges), class-centroid features + Gaussian noise."""
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    y = torch.randint(0, num_classes, (num_nodes,), generator=g)
    half = num_edges // 2
    u = torch.randint(0, num_nodes, (half,), generator=g)
    same = torch.rand(half, generator=g) < homophily
    # partner: a random node of the same class (homophilous) or any node
    order = torch.argsort(y, stable=True)
    counts = torch.bincount(y, minlength=num_classes)
    starts = torch.cumsum(counts, 0) - counts
    cu = y[u]
    pick = (torch.rand(half, generator=g) * counts[cu].clamp(min=1)).long().clamp(max=num_nodes - 1)
    v_same = order[(starts[cu] + pick).clamp(max=num_nodes - 1)]
    v_any = torch.randint(0, num_nodes, (half,), generator=g)
    v = torch.where(same, v_same, v_any)
    edge_index = torch.stack([torch.cat([u, v]), torch.cat([v, u])], dim=0)
    centroids = torch.randn(num_classes, num_features, generator=g)
    x = centroids[y] + noise * torch.randn(num_nodes, num_features, generator=g)
    return Data(x=x.to(device), edge_index=edge_index.to(device), y=y.to(device))


def _masks(data, n_train, n_val, n_test, seed):
    g = torch.Generator()
    g.manual_seed(seed + 1)
    n = data.num_nodes
    perm = torch.randperm(n, generator=g)
    for name, lo, hi in (('train_mask', 0, n_train), ('val_mask', n_train, n_train + n_val),
                         ('test_mask', n_train + n_val, n_train + n_val + n_test)):
        m = torch.zeros(n, dtype=torch.bool)
        m[perm[lo:hi]] = True
        setattr(data, name, m)
    return data


class SyntheticDataset:
    """dataset[0] -> Data; .num_features / .num_classes like a PyG dataset."""

    def __init__(self, datas, num_classes):
        self._datas = datas
        self.num_classes = num_classes

    def __getitem__(self, i):
        return self._datas[i]

    def __len__(self):
        return len(self._datas)

    def __iter__(self):
        return iter(self._datas)

    @property
    def num_features(self):
        return self._datas[0].num_features

    num_node_features = num_features


def _named(name, scale=1.0, seed=0, homophily=0.8, noise=2.0):
    n, e, f, c, tr, va, te = SHAPES[name]
    if scale != 1.0:
        n, e = max(int(n * scale), c * 4), max(int(e * scale), 16)
        tr, va, te = max(int(tr * scale), 1), max(int(va * scale), 1), max(int(te * scale), 1)
    data = planted_partition(n, e, f, c, homophily=homophily, noise=noise, seed=seed)
    _masks(data, tr, va, te, seed)
    return SyntheticDataset([data], c)


def Planetoid(root=None, name='Cora', scale=1.0, seed=0, **kwargs):
    return _named(name if name in SHAPES else name.capitalize(), scale, seed)


def Reddit(root=None, scale=1.0, seed=0, **kwargs):
    return _named('Reddit', scale, seed)


def Flickr(root=None, scale=1.0, seed=0, **kwargs):
    return _named('Flickr', scale, seed)


def PPI(root=None, split='train', seed=0, **kwargs):
    """20/2/2 graphs of ~2.4 k nodes, 50 features, 121 sigmoid labels."""
    n_graphs = {'train': 20, 'val': 2, 'test': 2}[split]
    base = {'train': 0, 'val': 100, 'test': 200}[split]
    datas = []
    for gidx in range(n_graphs):
        g = torch.Generator()
        g.manual_seed(seed + base + gidx)
        n = int(torch.randint(1800, 3200, (1,), generator=g))
        d = planted_partition(n, int(n * 28.7), 50, 8, homophily=0.7, noise=1.0, seed=seed + base + gidx)
        proj = torch.randn(50, 121, generator=g)
        d.y = ((d.x @ proj) > 0).float()
        datas.append(d)
    return SyntheticDataset(datas, 121)


def reddit_like_device(num_nodes=232965, num_edges=114615892, num_features=602, num_classes=41,
                       seed=0, device='cuda', sigma=1.0):
    """Reddit-SHAPED graph generated directly on the device: symmetric directed
    edge list with log-normal node popularity (mean in-degree num_edges/num_nodes), X ~ N(0,1),
    y ~ U{0..C-1}.  Endpoints are drawn by inverse-CDF sampling of the popularity weights."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    w = torch.exp(sigma * torch.randn(num_nodes, generator=g, device=device))
    cdf = torch.cumsum(w.double(), 0)
    cdf = (cdf / cdf[-1]).float()
    half = num_edges // 2
    u = torch.searchsorted(cdf, torch.rand(half, generator=g, device=device)).clamp_(max=num_nodes - 1)
    v = torch.searchsorted(cdf, torch.rand(half, generator=g, device=device)).clamp_(max=num_nodes - 1)
    edge_index = torch.stack([torch.cat([u, v]), torch.cat([v, u])], dim=0)
    del u, v, cdf, w
    x = torch.randn(num_nodes, num_features, generator=g, device=device)
    y = torch.randint(0, num_classes, (num_nodes,), generator=g, device=device)
    return Data(x=x, edge_index=edge_index, y=y)
