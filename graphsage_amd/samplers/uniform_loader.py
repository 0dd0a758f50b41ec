"""UniformLoader / NeighborSampler with the reference's API (reference
graphsage/samplers/uniform_loader.py:16-88, :91-252).

What differs from the reference (a vendored PyG NeighborLoader) is where the work runs:
  * the CSC is built ONCE per data object on the device (the reference re-sorts the edge list
    on the CPU for each of a trainer's three loaders: uniform_loader.py:32,
    trainers/node_level_graphsage_trainers.py:18-25) and stays resident in HBM;
  * sampling is a HIP kernel sequence on the current stream, not torch DataLoader worker
    processes: `num_workers` / `persistent_workers` are accepted and ignored;
  * batch.x is gathered on the device on first access (lazy), so model.forward_batch(batch)
    can read the global table directly and never materialise it.
Batches expose what the reference's callers use: x, y, n_id, edge_index ([src; dst], local
ids), batch_size, to(device); the first batch_size rows are the seeds in input order.
"""
from typing import Callable, List, Optional, Sequence

import torch
from torch import Tensor

from .. import functional as SF
from .. import ops, settings
from ..data import Data
from ..dist import batch_counter, shard_seeds


def _device_for(data):
    dev = settings.DEVICE
    if dev.type != 'cuda':
        raise RuntimeError("graphsage_amd samples on the GPU: no CUDA/HIP device is available")
    return dev


class DeviceGraph:
    """Device-resident CSC + feature/label tables of one data object (built once, shared by
    every loader over it).  Feature rows are padded to a multiple of 16 floats (602 -> 608) so
    that every row starts on a 64-byte boundary and 16-byte vector loads are legal."""

    def __init__(self, data, device):
        edge_index = data.edge_index.to(device)
        self.num_nodes = int(data.num_nodes)
        self.colptr, self.row, self.perm = ops.csc_build(edge_index, self.num_nodes)
        self.device = device
        self._x_src = None
        self._x_dev = None
        self._y_src = None
        self._y_dev = None

    def features(self, data):
        x = data.x
        if self._x_dev is None or self._x_src is not x:
            F = x.size(1)
            ld = (F + 15) // 16 * 16
            buf = torch.zeros((x.size(0), ld), dtype=torch.float32, device=self.device)
            buf[:, :F] = x.to(self.device, torch.float32)
            self._x_dev = buf[:, :F]  # a [N, F] view whose rows are 64-byte aligned (stride ld)
            self._x_src = x
        return self._x_dev

    def labels(self, data):
        y = getattr(data, 'y', None)
        if y is None:
            return None
        if self._y_dev is None or self._y_src is not y:
            self._y_dev = y.to(self.device).contiguous()
            self._y_src = y
        return self._y_dev


def device_graph(data):
    g = data.__dict__.get('_sage_graph')
    dev = _device_for(data)
    if g is None or g.num_nodes != data.num_nodes or g.device != dev:
        g = DeviceGraph(data, dev)
        data.__dict__['_sage_graph'] = g
    return g


class NeighborSampler:
    """reference uniform_loader.py:16-88 (homogeneous branch).  __call__(index) returns
    (node, row, col, edge, batch_size) exactly like the reference's collate function."""

    def __init__(self, data, num_neighbors, replace: bool = False, directed: bool = True,
                 input_node_type: Optional[str] = None, seed: Optional[int] = None):
        if not isinstance(data, Data):
            raise TypeError(f'NeighborLoader found invalid type: {type(data)}')
        if replace:
            raise NotImplementedError("replace=True is not supported (never used by the reference's scripts)")
        if not directed:
            raise NotImplementedError("directed=False is not supported (never used by the reference's scripts)")
        assert isinstance(num_neighbors, (list, tuple))
        self.data_cls = data.__class__
        self.replace, self.directed = replace, directed
        self._attach(device_graph(data), num_neighbors, seed)

    @classmethod
    def from_graph(cls, graph, num_neighbors, seed: Optional[int] = None):
        """Sampler over an already resident DeviceGraph (no edge list needed any more)."""
        self = cls.__new__(cls)
        self.data_cls = Data
        self.replace, self.directed = False, True
        self._attach(graph, num_neighbors, seed)
        return self

    def _attach(self, graph, num_neighbors, seed):
        self.num_neighbors = [int(k) for k in num_neighbors]
        self.graph = graph
        self.colptr, self.row, self.perm = self.graph.colptr, self.graph.row, self.graph.perm
        self.seed = settings.SEED if seed is None else int(seed)
        self.counter = 0          # advances once per sampled batch
        self._stamp = 0
        self._table = ops.new_relabel_table(self.graph.num_nodes, self.graph.device)
        self._bufs = {}

    def capacities(self, batch_size):
        """Upper bounds on nodes/edges of a batch; never exceeded for non-negative fan-outs.
        A fan-out of -1 (all neighbours) is bounded by the edge count of the graph; if a
        batch exceeds it (duplicate seeds), the sampler reports it and __call__ raises."""
        n = self.graph.num_nodes
        e_total = self.row.numel()
        frontier, nodes, edges = batch_size, batch_size, 0
        for k in self.num_neighbors:
            hop_edges = frontier * k if k >= 0 else e_total
            edges += hop_edges
            frontier = min(hop_edges, n)
            nodes += frontier
        return min(nodes, n + batch_size), min(edges, 2 ** 31 - 2)

    def buffers(self, batch_size):
        buf = self._bufs.get(batch_size)
        if buf is None:
            node_cap, edge_cap = self.capacities(batch_size)
            buf = ops.SampleBuffers(self.graph.num_nodes, node_cap, edge_cap, self.graph.device)
            self._bufs[batch_size] = buf
        return buf

    def sample_async(self, index: Tensor, counter: Optional[int] = None):
        """Enqueue the sampler for `index` (device int64); returns the buffers (no sync)."""
        if counter is None:
            counter = self.counter
            self.counter += 1
        buf = self.buffers(index.numel())
        self._stamp += 1
        if self._stamp >= 2 ** 32 - 1:  # stamps exhausted: start over with a clean table
            self._table.fill_(-1)
            self._stamp = 1
        ops.sample_fanout(self.colptr, self.row, index, self.num_neighbors, self.seed, counter,
                          self._table, self._stamp, buf)
        return buf

    def __call__(self, index):
        if not isinstance(index, Tensor):
            index = torch.tensor(index, dtype=torch.int64)
        index = index.to(self.graph.device, torch.int64).contiguous()
        buf = self.sample_async(index)
        state = buf.state.tolist()  # one D2H sync per batch: the sizes
        if state[ops.ST_ERROR] != 0:
            raise RuntimeError(f"neighbor sampling failed (device error bits {state[ops.ST_ERROR]:#x}): "
                               "seed out of range or capacity exceeded")
        n, e = state[ops.ST_NODES], state[ops.ST_EDGES]
        self.last_state = state
        self.last_seg_ptr = buf.seg_ptr[:n + 1]
        return buf.node[:n].clone(), buf.src[:e].clone(), buf.dst[:e].clone(), buf.eid[:e].clone(), index.numel()


class Batch(Data):
    """A sampled mini-batch.  x / y are gathered on first access."""

    def __init__(self, graph, data, node, edge_index, batch_size, feat_table, label_table):
        super().__init__()
        self.__dict__['_graph'] = graph
        self.__dict__['_feat_table'] = feat_table
        self.__dict__['_label_table'] = label_table
        self.n_id = node
        self.edge_index = edge_index
        self.batch_size = batch_size

    def __getattr__(self, name):
        # only called when normal lookup fails: lazy node-level attributes
        if name == 'x':
            t = self.__dict__.get('_feat_table')
            if t is None:
                raise AttributeError(name)
            v = ops.gather_rows(t, self.__dict__['n_id'], checked=True)
            self.__dict__['x'] = v
            return v
        if name == 'y':
            t = self.__dict__.get('_label_table')
            if t is None:
                raise AttributeError(name)
            v = ops.gather_rows(t, self.__dict__['n_id'], checked=True)
            self.__dict__['y'] = v
            return v
        raise AttributeError(name)

    @property
    def num_nodes(self):
        return self.__dict__['n_id'].numel()


class _IndexDataset(Sequence):
    def __init__(self, idx):
        self.idx = idx

    def __len__(self):
        return len(self.idx)

    def __getitem__(self, i):
        return self.idx[i]


def get_input_node_indices(data, input_nodes):
    """reference uniform_loader.py:267-287 (homogeneous branch) -> 1-D int64 CPU tensor."""
    if input_nodes is None:
        return torch.arange(data.num_nodes, dtype=torch.int64)
    if isinstance(input_nodes, Tensor):
        if input_nodes.dtype == torch.bool:
            return input_nodes.nonzero(as_tuple=False).view(-1).cpu()
        return input_nodes.to(torch.int64).view(-1).cpu()
    return torch.tensor(list(input_nodes), dtype=torch.int64)


class NeighborLoader:
    """reference uniform_loader.py:91-252.  Iterable of Batch; len() = number of batches;
    `.dataset` has len() = number of input nodes (used by the reference's progress bars)."""

    def __init__(self, data, num_neighbors: List[int], input_nodes=None, replace: bool = False,
                 directed: bool = True, transform: Callable = None,
                 neighbor_sampler: Optional[NeighborSampler] = None, batch_size: int = 1,
                 shuffle: bool = False, drop_last: bool = False, num_workers: int = 0,
                 persistent_workers: bool = False, seed: Optional[int] = None,
                 rank: int = 0, world_size: int = 1, **kwargs):
        kwargs.pop('dataset', None)
        kwargs.pop('collate_fn', None)
        self.data = data
        self.num_neighbors = num_neighbors
        self.input_nodes = input_nodes
        self.replace, self.directed, self.transform = replace, directed, transform
        self.data.n_id = torch.arange(data.num_nodes)  # reference uniform_loader.py:224
        self.neighbor_sampler = neighbor_sampler or NeighborSampler(data, num_neighbors, replace,
                                                                    directed, seed=seed)
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), shuffle, drop_last
        self.rank, self.world_size = int(rank), int(world_size)
        idx = get_input_node_indices(data, input_nodes)
        idx = shard_seeds(idx, self.rank, self.world_size)   # data-parallel shard
        self.dataset = _IndexDataset(idx)
        self._idx_dev = idx.to(self.neighbor_sampler.graph.device)
        self.epoch = 0
        self.generator = torch.Generator()
        self.generator.manual_seed(self.neighbor_sampler.seed)

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _order(self):
        if not self.shuffle:
            return self._idx_dev
        perm = torch.randperm(len(self.dataset), generator=self.generator)
        return self._idx_dev[perm.to(self._idx_dev.device)]

    def make_batch(self, out):
        """transform_fn (reference uniform_loader.py:235-249) on device."""
        node, row, col, edge, batch_size = out
        sampler = self.neighbor_sampler
        graph = sampler.graph
        edge_index = torch.stack([row, col], dim=0)
        st = sampler.last_state
        hops = len(sampler.num_neighbors)
        plan = SF.EdgePlan(sampler.last_seg_ptr.clone(), row, node.numel(), node.numel(),
                           hop_nodes=[st[ops.ST_HOP_NODES + h] for h in range(hops + 1)],
                           hop_edges=[st[ops.ST_HOP_EDGES + h] for h in range(hops + 1)],
                           node=node, table=graph.features(self.data) if hasattr(self.data, 'x') else None)
        edge_index._sage_plan = plan
        batch = Batch(graph, self.data, node, edge_index, batch_size,
                      plan.table, graph.labels(self.data))
        return batch if self.transform is None else self.transform(batch)

    def __iter__(self):
        order = self._order()
        n = order.numel()
        nb = len(self)
        sampler = self.neighbor_sampler
        for b in range(nb):
            index = order[b * self.batch_size: min((b + 1) * self.batch_size, n)].contiguous()
            # counter = f(epoch, batch, rank): distinct Philox streams for every batch of every rank
            counter = batch_counter(self.epoch, nb, b, self.rank, self.world_size)
            index_dev = index
            buf = sampler.sample_async(index_dev, counter=counter)
            state = buf.state.tolist()
            if state[ops.ST_ERROR] != 0:
                raise RuntimeError(f"neighbor sampling failed (device error bits {state[ops.ST_ERROR]:#x})")
            nn, ne = state[ops.ST_NODES], state[ops.ST_EDGES]
            sampler.last_state = state
            sampler.last_seg_ptr = buf.seg_ptr[:nn + 1]
            out = (buf.node[:nn].clone(), buf.src[:ne].clone(), buf.dst[:ne].clone(),
                   buf.eid[:ne].clone(), index.numel())
            yield self.make_batch(out)
        self.epoch += 1

    def __repr__(self):
        return f'{self.__class__.__name__}()'
