"""Minimal stand-in for torch_geometric.data.Data (absent from this image): an attribute bag
holding node-level tensors (first dim = num_nodes), edge_index and anything else.
Only what the hot path's callers touch (reference samplers/uniform_loader.py:224,235-249;
trainers/node_level_graphsage_trainers.py:37-40)."""
import torch


class Data:
    def __init__(self, x=None, edge_index=None, y=None, **kwargs):
        if x is not None:
            self.x = x
        if edge_index is not None:
            self.edge_index = edge_index
        if y is not None:
            self.y = y
        for k, v in kwargs.items():
            setattr(self, k, v)

    # -- torch_geometric.data.Data surface used by the reference's callers
    @property
    def num_nodes(self):
        if "num_nodes_" in self.__dict__:
            return self.__dict__["num_nodes_"]
        for key in ("x", "y", "n_id"):
            v = self.__dict__.get(key)
            if isinstance(v, torch.Tensor):
                return v.size(0)
        ei = self.__dict__.get("edge_index")
        return int(ei.max()) + 1 if ei is not None and ei.numel() else 0

    @num_nodes.setter
    def num_nodes(self, n):
        self.__dict__["num_nodes_"] = int(n)

    @property
    def num_edges(self):
        ei = self.__dict__.get("edge_index")
        return ei.size(1) if ei is not None else 0

    @property
    def num_features(self):
        x = self.__dict__.get("x")
        return x.size(1) if x is not None and x.dim() == 2 else 0

    num_node_features = num_features

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]

    def __contains__(self, key):
        return key in self.__dict__

    def __getitem__(self, key):
        return getattr(self, key)

    def to(self, device, *keys, non_blocking=False):
        for k, v in list(self.__dict__.items()):
            if k.startswith("_") or (keys and k not in keys):
                continue
            if isinstance(v, torch.Tensor):
                moved = v.to(device, non_blocking=non_blocking)
                # keep host-side annotations (the sampler's plan) attached to the tensor
                if moved is not v and hasattr(v, "_sage_plan"):
                    moved._sage_plan = v._sage_plan
                self.__dict__[k] = moved
        return self

    def cpu(self):
        return self.to("cpu")

    def cuda(self):
        return self.to("cuda")

    def __repr__(self):
        parts = []
        for k in self.keys():
            v = self.__dict__[k]
            parts.append(f"{k}={list(v.shape)}" if isinstance(v, torch.Tensor) else f"{k}={v}")
        return f"{self.__class__.__name__}({', '.join(parts)})"


def collate_graphs(graphs):
    """Several graphs as ONE disconnected graph (what torch_geometric's DataLoader hands to
    reference trainers/graph_level_graphsage_trainers.py:29-41): node tensors concatenated,
    edge_index shifted by the node offsets; `batch` maps nodes to graphs, `num_graphs` counts them."""
    graphs = list(graphs)
    out = Data()
    offset, eis, owner = 0, [], []
    for g, d in enumerate(graphs):
        eis.append(d.edge_index + offset)
        owner.append(torch.full((d.num_nodes,), g, dtype=torch.int64))
        offset += d.num_nodes
    for key in graphs[0].keys():
        vals = [getattr(d, key) for d in graphs]
        if key == "edge_index" or not all(isinstance(v, torch.Tensor) for v in vals):
            continue
        if all(v.dim() >= 1 and v.size(0) == d.num_nodes for v, d in zip(vals, graphs)):
            setattr(out, key, torch.cat(vals, dim=0))
    out.edge_index = torch.cat(eis, dim=1)
    out.batch = torch.cat(owner)
    out.num_graphs = len(graphs)
    return out


class GraphDataLoader:
    """Iterable of collated graph batches with the two attributes the reference's graph-level
    trainers read: len(loader.dataset) and batch.num_graphs (the role torch_geometric.loader.
    DataLoader plays in examples/ppi_graphsage_supervised.py:32-34; absent from this image)."""

    def __init__(self, dataset, batch_size=1, shuffle=False, seed=0, **unused):
        self.dataset = dataset
        self.batch_size, self.shuffle = int(batch_size), bool(shuffle)
        self.generator = torch.Generator().manual_seed(seed)

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        for b in range(0, n, self.batch_size):
            yield collate_graphs(self.dataset[i] for i in order[b:b + self.batch_size])
