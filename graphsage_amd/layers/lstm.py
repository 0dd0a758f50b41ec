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
above with torch.nn.functional on the CPU, and at the Reddit shape against torch.nn.LSTMCell.

Kernels: x_flat [T, n_in*F] materialised once per call (sage_lstm_gather_flat: slot rule and, for
layer 0, the batch's node vector applied inside the kernel; 0.8 GB at the Reddit shape — nothing
in 288 GB of HBM) -> gates = x_flat @ W_ih^T on the MFMA GEMM, ONE launch per gate block over
K = n_in*F with both operands contiguous in k -> zero-state cell non-linearity (HIP, biases added
on the way in).  Round 2 ran the contraction as n_in K-segments through per-slot row maps (13
two-segment launches that each read-modify-write the gates) and re-laid W_ih into a padded copy
on every call; the static-shape engine's route (graphsage_amd/engine.py) replaced
both, and this module path now calls the same pieces.  W_ih is read in place when its rows are
16-byte aligned (a model whose parameters the engine flattened), else through ONE padded copy
(`ops.aligned_rows`).  The forget gate multiplies the zero initial cell state, so its quarter of
W_ih is neither multiplied nor given a gradient (3/4 of the flops); W_hh multiplies the zero hidden
state and receives no gradient, exactly as in the reference's single-step use.
"""
import torch
from torch.autograd import Function

from .. import ops


def _blocks(fdim):
    """Row blocks of W_ih (torch order i, f, g, o) that reach the output from a zero state."""
    return ((0, fdim), (2 * fdim, 4 * fdim))


class _LstmGatesFn(Function):
    """gates[t] = x_flat[t] @ W_ih^T   (rows of the f gate: zeros), nonempty[t] = target t has an in-edge."""

    @staticmethod
    def forward(ctx, x, w_ih, seg_ptr, src, node_map, num_targets, n_in):
        T, fdim = int(num_targets), x.size(1)
        x = ops.aligned_rows(x)
        x_flat = ops.empty_rows(T, n_in * fdim, x.device)
        nonempty = torch.empty(T, dtype=torch.float32, device=x.device)
        gates = ops.empty_rows(T, 4 * fdim, x.device, zero=True)
        w = ops.aligned_rows(w_ih.detach())
        if T > 0:
            ops.lstm_gather_flat(x, seg_ptr, src, T, n_in, x_flat, nonempty, node_map=node_map)
            for r0, r1 in _blocks(fdim):
                ops.linear_concat_fwd(x_flat, w[r0:r1], m=T, out=gates[:, r0:r1])
        ctx.save_for_backward(x_flat, w, seg_ptr, src)
        ctx.fdim, ctx.n_in, ctx.n_rows, ctx.mapped = fdim, int(n_in), x.size(0), node_map is not None
        ctx.mark_non_differentiable(nonempty)
        return gates, nonempty

    @staticmethod
    def backward(ctx, dgates, _dnonempty):
        x_flat, w, seg_ptr, src = ctx.saved_tensors
        fdim, n_in, T = ctx.fdim, ctx.n_in, x_flat.size(0)
        dgates = ops.aligned_rows(dgates.contiguous())
        dx = dw = None
        if ctx.needs_input_grad[1]:
            dw = ops.empty_rows(4 * fdim, n_in * fdim, x_flat.device, zero=True)
            if T > 0:
                for r0, r1 in _blocks(fdim):
                    ops.linear_dweight(dgates[:, r0:r1], x_flat, m=T, dw=dw[r0:r1])
        if ctx.needs_input_grad[0]:
            if ctx.mapped:
                raise RuntimeError("gradient w.r.t. a node_map'ed feature table is not supported")
            dx = ops.empty_rows(ctx.n_rows, fdim, x_flat.device, zero=True)
            if T > 0:
                (a0, a1), (b0, b1) = _blocks(fdim)
                d_flat = ops.empty_rows(T, n_in * fdim, x_flat.device)
                ops.linear_dinput(dgates[:, a0:a1], w[a0:a1], m=T, out=d_flat)
                ops.linear_dinput(dgates[:, b0:b1], w[b0:b1], m=T, out=d_flat, accumulate=True)
                ops.lstm_scatter_flat(d_flat, seg_ptr, src, T, n_in, dx)
        return dx, dw, None, None, None, None, None


class _LstmCell0Fn(Function):
    """h = cell(gates + b_ih + b_hh) from a zero state; the biases' gradient = column sums of d gates."""

    @staticmethod
    def forward(ctx, gates, nonempty, b_ih, b_hh, num_features):
        ctx.save_for_backward(gates, nonempty, b_ih, b_hh)
        ctx.nf = int(num_features)
        return ops.lstm_cell0_fwd(gates, nonempty, num_features, bias_a=b_ih, bias_b=b_hh)

    @staticmethod
    def backward(ctx, dh):
        gates, nonempty, b_ih, b_hh = ctx.saved_tensors
        dgates = ops.lstm_cell0_bwd(gates, nonempty, dh.contiguous(), ctx.nf, bias_a=b_ih, bias_b=b_hh)
        dbias = dgates.sum(0) if (ctx.needs_input_grad[2] or ctx.needs_input_grad[3]) else None
        return dgates, None, dbias, dbias, None


def lstm_aggregate(layer, x_src, plan, num_targets, node_map=None):
    n_in = layer.lstm_n_inputs
    fdim = x_src.size(1)
    if node_map is not None and x_src.requires_grad:
        raise RuntimeError("gradient w.r.t. a node_map'ed feature table is not supported")
    gates, nonempty = _LstmGatesFn.apply(x_src, layer.lstm.weight_ih, plan.seg_ptr, plan.src, node_map,
                                         num_targets, n_in)
    return _LstmCell0Fn.apply(gates, nonempty, layer.lstm.bias_ih, layer.lstm.bias_hh, fdim)
