"""The 'lstm' aggregator with the reference's parameters (`lstm = LSTMCell(n_in*F, F)`,
reference graphsage/layers/sage.py:67-69) and the computation its aggregate() is after
(:147-193), made well-defined:

  for every TARGET node t with at least one in-edge:
      rows  = features of its neighbours in edge order, the first n_in kept      (:154-159; the
              reference truncates with random.sample — unseeded, not reproducible)
      rows  = rows padded to n_in by repeating the first row                     (:164)
      x_t   = rows flattened to one vector of n_in*F values                      (:179)
      h_t,_ = LSTMCell(x_t) from a zero (h, c) state; the aggregate is h_t        (:187)
  targets without in-edges aggregate to 0                                         (:121-122)

**Parity unpinned, by construction**: the reference's implementation is disabled upstream as
broken (settings.py:48-50) — it keys its dict by SOURCE ids, indexes the per-edge tensor with
node ids (:164) and returns rows in set-iteration order (:193) — so there is no reference output
to match; tests compare against oracle/sage_ref.lstm_aggregate_ref, which restates the list
above with torch.nn.functional on the CPU.

Kernels: slot indices (HIP) -> row gather (HIP) -> gates = x_flat @ W_ih^T + b_ih + b_hh on the
fp32 MFMA GEMM -> zero-state cell non-linearity (HIP).  W_hh multiplies the zero hidden state
and receives a zero gradient, exactly as in the reference's single-step use.
"""
import torch
from torch.autograd import Function

from .. import functional as SF
from .. import ops


class _GatherRowsFn(Function):
    """x[idx] with a scatter-add backward (kernel: gather_reduce_bwd over unit segments)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.xshape = tuple(x.shape)
        return ops.gather_rows(x if x.stride(-1) == 1 else x.contiguous(), idx, checked=True)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None
        dx = torch.zeros(ctx.xshape, dtype=torch.float32, device=dout.device)
        unit = torch.arange(idx.numel() + 1, dtype=torch.int64, device=dout.device)
        ops.gather_reduce_bwd(dout.contiguous(), unit, idx, idx.numel(), "sum", dx)
        return dx, None


class _LstmCell0Fn(Function):
    @staticmethod
    def forward(ctx, gates, nonempty, num_features):
        gates = gates.contiguous()
        ctx.save_for_backward(gates, nonempty)
        ctx.nf = int(num_features)
        return ops.lstm_cell0_fwd(gates, nonempty, num_features)

    @staticmethod
    def backward(ctx, dh):
        gates, nonempty = ctx.saved_tensors
        return ops.lstm_cell0_bwd(gates, nonempty, dh.contiguous(), ctx.nf), None, None


def lstm_aggregate(layer, x_src, plan, num_targets, node_map=None):
    n_in = layer.lstm_n_inputs
    fdim = x_src.size(1)
    slots, nonempty = ops.segment_slots(plan.seg_ptr, plan.src, num_targets, n_in, node_map=node_map)
    rows = _GatherRowsFn.apply(x_src, slots)                       # [T*n_in, F]
    x_flat = rows.view(num_targets, n_in * fdim)
    bias = layer.lstm.bias_ih + layer.lstm.bias_hh
    gates = SF.linear_concat(x_flat, layer.lstm.weight_ih, bias=bias)
    return _LstmCell0Fn.apply(gates, nonempty, fdim)
