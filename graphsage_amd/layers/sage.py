"""SAGE layer with the reference's module API (reference graphsage/layers/sage.py:17-134):

    SAGE(in_channels, out_channels, aggregator='mean', normalize=True, root_weight=True)
    forward(x | (x_src, x_dst), edge_index[2,E] | adj_t (CSR), size=None) -> [N, out_channels]

Parameter names/shapes match (`lin_l.weight`, `lin_r.weight` bias-free [out,in];
`pool.weight/bias` [in,in]; `lstm.*` — :67-81) so state_dicts interchange.  The arithmetic
is three HIP kernels: fused gather+segment-reduce, the self||neighbour Linear on MFMA, and
L2-normalise(+ReLU); message()/aggregate()/propagate() of PyG's MessagePassing have no
counterpart because x_j is never materialised.
"""
from typing import Optional, Tuple, Union

import torch
from torch import Tensor

from .. import functional as SF
from .. import settings

AGGREGATORS = ['mean', 'max', 'sum', 'gcn', 'lstm', 'bilstm', 'max_pool', 'mean_pool', None]
_REDUCE = {'mean': 'mean', 'max': 'max', 'sum': 'sum', 'gcn': 'mean', 'max_pool': 'max',
           'mean_pool': 'mean'}


class SAGE(torch.nn.Module):
    def __init__(self, in_channels: Union[int, Tuple[int, int]], out_channels: int,
                 aggregator: str = 'mean', normalize: bool = True, root_weight: bool = True,
                 **kwargs):
        super().__init__()
        # PyG MessagePassing kwargs the reference forwards (aggr=, flow=, node_dim=): `aggr` is
        # also assigned after construction by examples/others/citation_graphsaint_supervised.py:36-39
        self.aggr = kwargs.pop('aggr', 'add')
        self.flow = kwargs.pop('flow', 'source_to_target')
        self.node_dim = kwargs.pop('node_dim', -2)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.normalize = normalize
        self.root_weight = root_weight
        self.aggregator = aggregator
        assert self.aggregator in AGGREGATORS
        if self.aggregator == 'bilstm':
            # disabled upstream as broken (reference settings.py:48-50); out of scope
            raise NotImplementedError("aggregator 'bilstm' is not supported")
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        if self.aggregator == 'gcn':
            self.root_weight = False  # reference layers/sage.py:62-65
        if self.aggregator == 'lstm':
            self.lstm_n_inputs = settings.LSTM_NUM_INPUTS
            self.lstm = torch.nn.LSTMCell(self.lstm_n_inputs * in_channels[0], in_channels[0])
        if self.aggregator in {'max_pool', 'mean_pool'}:
            self.pool = torch.nn.Linear(in_channels[0], in_channels[0], bias=True)
        self.lin_l = torch.nn.Linear(in_channels[0], out_channels, bias=False)  # neighbours
        if self.root_weight:
            self.lin_r = torch.nn.Linear(in_channels[1], out_channels, bias=False)  # root
        self.reset_parameters()

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        if self.root_weight:
            self.lin_r.reset_parameters()
        if self.aggregator == 'lstm':
            self.lstm.reset_parameters()

    # ------------------------------------------------------------------ public forward
    def forward(self, x: Union[Tensor, Tuple[Tensor, Optional[Tensor]]], edge_index: Tensor,
                size=None) -> Tensor:
        # `size` is sometimes handed an edge_weight by callers (reference
        # examples/others/citation_graphsaint_supervised.py:42); like the reference, ignore it.
        if isinstance(x, Tensor):
            x = (x, x)
        x_src, x_dst = x
        dense = isinstance(edge_index, Tensor) and edge_index.layout == torch.strided
        if x_dst is not None:
            num_dst = x_dst.size(0)
        elif isinstance(size, (tuple, list)) and size[1] is not None:
            num_dst = int(size[1])
        elif not dense:
            # adj_t (CSR, row = target): the message_and_aggregate form, reference :213-232
            num_dst = SF.plan_for(edge_index, x_src.size(0), 0).num_dst
        else:
            num_dst = int(edge_index[1].max()) + 1 if edge_index.numel() else 0
        # (gcn's self-loops, reference layers/sage.py:114-117, are applied arithmetically in
        # apply_plan: the edge list is not rewritten)
        plan = SF.plan_for(edge_index, x_src.size(0), num_dst)
        return self.apply_plan(x_src, x_dst, plan, num_dst, relu=False)

    # ------------------------------------------------------------------ shared body
    def apply_plan(self, x_src, x_dst, plan, num_targets, relu=False, node_map=None,
                   num_src_rows=None):
        """Layer body on the first `num_targets` targets of `plan`.

        node_map: the batch's `node` vector when x_src/x_dst are the GLOBAL feature table
        (rows are then read through it; nothing is materialised).  num_src_rows: how many
        leading source rows the targets can reference (pruned pool GEMM)."""
        if self.aggregator is None:
            raise UnboundLocalError("aggregator=None leaves `reduce` unbound (reference layers/sage.py:209)")
        if self.aggregator == 'lstm':
            from .lstm import lstm_aggregate
            agg = lstm_aggregate(self, x_src, plan, num_targets, node_map)
        else:
            feats = x_src
            fmap = node_map
            if self.aggregator in {'max_pool', 'mean_pool'}:
                # relu(pool(x_j)) is row-wise, so apply it once per SOURCE NODE instead of once
                # per edge (reference layers/sage.py:201-207): identical values, E/N times fewer flops
                n_rows = num_src_rows if num_src_rows is not None else (
                    node_map.numel() if node_map is not None else x_src.size(0))
                feats = SF.linear_concat(x_src, self.pool.weight, bias=self.pool.bias, relu=True,
                                         rows1=node_map, m=n_rows)
                fmap = None
            if self.aggregator == 'gcn':
                # mean over the neighbours and ONE self-loop for every target that has an in-edge
                # (reference layers/sage.py:114-117 appends the loops to edge_index and scatters):
                # (sum_j x_j + x_t) / (deg_t + 1) if deg_t > 0 else 0 — same values, same order of
                # summation (the loops are appended last), inside the gather kernel
                # (SAGE_REDUCE_GCN): no edge list rewrite, prunable
                agg = SF.gather_reduce(feats, plan.seg_ptr, plan.src, num_targets, 'gcn', node_map=fmap)
            else:
                agg = SF.gather_reduce(feats, plan.seg_ptr, plan.src, num_targets,
                                       _REDUCE[self.aggregator], node_map=fmap)
        if self.root_weight and x_dst is not None:
            z = SF.linear_concat(agg, self.lin_l.weight, x_dst, self.lin_r.weight,
                                 rows2=node_map, m=num_targets)
        else:
            z = SF.linear_concat(agg, self.lin_l.weight, m=num_targets)
        if self.normalize:
            return SF.l2norm_relu(z, relu)
        return z.relu() if relu else z

    def __repr__(self):
        return (f'{self.__class__.__name__}({self.in_channels}, {self.out_channels}, '
                f'aggregator={self.aggregator})')
