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
