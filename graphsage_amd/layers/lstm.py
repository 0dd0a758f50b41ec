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

Kernels: slot indices (HIP) -> gates = x_flat @ W_ih^T + b_ih + b_hh on the MFMA GEMM -> zero-state
cell non-linearity (HIP).  x_flat [T, n_in*F] (0.8 GB at the Reddit shape) is never materialised:
the contraction runs as n_in K-segments of width F, segment j reading the feature rows through
its own row map slots[j] (two segments per launch, `out +=`); small target sets (the layers above
layer 0: T = batch size) materialise x_flat and contract over all slots at once.  The forget gate multiplies the
zero initial cell state, so its quarter of W_ih is neither multiplied nor given a gradient
(3/4 of the flops); W_hh multiplies the zero hidden state and receives a zero gradient, exactly
as in the reference's single-step use.
"""
import torch
from torch.autograd import Function

from .. import ops


def _blocks(fdim):
    """Row blocks of W_ih (torch order i, f, g, o) that reach the output from a zero state."""
    return ((0, fdim), (2 * fdim, 4 * fdim))


_FLAT_BYTES = 256 << 20      # x_flat is materialised up to this size (the layers above layer 0: T = batch size)


class _LstmGatesFlatFn(Function):
    """The same gates for a SMALL target set (T x n_in x F floats within _FLAT_BYTES, F a multiple
    of 16 so that every slot's columns are 64-byte aligned): x_flat [T, n_in*F] is gathered once and
    the contraction over all slots is one launch per row block of W_ih — at T = 512 the per-slot
    K-segment launches are 16-workgroup grids of 11-13 us each, 2 x 26 of them per step."""

    @staticmethod
    def forward(ctx, x, slots, w_ih, bias):
        n_in, T = slots.shape
        fdim = x.size(1)
        idx = slots.t().reshape(-1)                      # row t*n_in + j = slot j of target t
        x_flat = x.index_select(0, idx).view(T, n_in * fdim)
        w = w_ih.detach()
        gates = bias.unsqueeze(0).expand(T, 4 * fdim).contiguous()
        for r0, r1 in _blocks(fdim):
            ops.linear_concat_fwd(x_flat, w[r0:r1], m=T, out=gates[:, r0:r1], accumulate=True)
        ctx.save_for_backward(x_flat, idx, w)
        ctx.fdim, ctx.n_rows = fdim, x.size(0)
        return gates

    @staticmethod
    def backward(ctx, dgates):
        x_flat, idx, w = ctx.saved_tensors
        fdim = ctx.fdim
        T = x_flat.size(0)
        dgates = dgates.contiguous()
        dx = dw = dbias = None
        if ctx.needs_input_grad[3]:
            dbias = dgates.sum(0)
        if ctx.needs_input_grad[2]:
            dw = torch.zeros_like(w)
            for r0, r1 in _blocks(fdim):
                ops.linear_dweight(dgates[:, r0:r1], x_flat, m=T, dw=dw[r0:r1])
        if ctx.needs_input_grad[0]:
            (a0, a1), (b0, b1) = _blocks(fdim)
            tmp = torch.empty_like(x_flat)
            ops.linear_dinput(dgates[:, a0:a1], w[a0:a1], m=T, out=tmp)
            ops.linear_dinput(dgates[:, b0:b1], w[b0:b1], m=T, out=tmp, accumulate=True)
            dx = torch.zeros((ctx.n_rows, fdim), dtype=torch.float32, device=x_flat.device)
            dx.index_add_(0, idx, tmp.view(-1, fdim))
        return dx, None, dw, dbias


class _LstmGatesFn(Function):
    """gates[t] = sum_j x[slots[j, t]] @ W_ih[:, jF:(j+1)F]^T + bias   (rows of the f gate: bias only)."""

    @staticmethod
    def forward(ctx, x, slots, w_ih, bias):
        n_in, T = slots.shape
        fdim = x.size(1)
        x = ops.aligned_rows(x)
        # W_ih [4F, n_in*F] re-laid as [4F, n_in, ld] with 64-byte aligned segments (F = 602: ld 608)
        ld = (fdim + 15) // 16 * 16
        w_pad = torch.zeros((4 * fdim, n_in, ld), dtype=torch.float32, device=x.device)
        w_pad[:, :, :fdim].copy_(w_ih.view(4 * fdim, n_in, fdim))
        gates = torch.empty((T, 4 * fdim), dtype=torch.float32, device=x.device)
        gates.copy_(bias.unsqueeze(0).expand(T, 4 * fdim))
        if T > 0:
            for j in range(0, n_in, 2):
                two = j + 1 < n_in
                for r0, r1 in _blocks(fdim):
                    ops.linear_concat_fwd(x, w_pad[r0:r1, j, :fdim], x if two else None,
                                          w_pad[r0:r1, j + 1, :fdim] if two else None,
                                          rows1=slots[j], rows2=slots[j + 1] if two else None, m=T,
                                          out=gates[:, r0:r1], accumulate=True)
        ctx.save_for_backward(x, slots, w_pad)
        ctx.fdim = fdim
        return gates

    @staticmethod
    def backward(ctx, dgates):
        x, slots, w_pad = ctx.saved_tensors
        n_in, T = slots.shape
        fdim = ctx.fdim
        dgates = dgates.contiguous()
        dx = dw = dbias = None
        if ctx.needs_input_grad[3]:
            dbias = dgates.sum(0)
        if ctx.needs_input_grad[2]:
            dw_pad = torch.zeros_like(w_pad)
            if T > 0:
                for j in range(0, n_in, 2):
                    for r0, r1 in _blocks(fdim):
                        g = dgates[:, r0:r1]
                        if j + 1 < n_in:
                            ops.linear_dweight2(g, x, dw_pad[r0:r1, j, :fdim], x, dw_pad[r0:r1, j + 1, :fdim],
                                                m=T, rows1=slots[j], rows2=slots[j + 1])
                        else:
                            ops.linear_dweight(g, x, m=T, rows=slots[j], dw=dw_pad[r0:r1, j, :fdim])
            dw = dw_pad[:, :, :fdim].reshape(4 * fdim, n_in * fdim)
        if ctx.needs_input_grad[0]:
            dx = ops.empty_rows(x.size(0), fdim, x.device, zero=True)
            if T > 0:
                # d x_flat [T, n_in * ld] in TWO contractions over all slots at once (the padded
                # W_ih read as [4F, n_in * ld]: its pad columns are zeros), then one scatter per slot;
                # per-slot contractions are 2 n_in launches of a 16-workgroup grid each (22 us apiece
                # at T = 512)
                ld = w_pad.size(2)
                w_flat = w_pad.view(4 * fdim, n_in * ld)
                tmp = ops.empty_rows(T, n_in * ld, x.device)
                (a0, a1), (b0, b1) = _blocks(fdim)
                ops.linear_dinput(dgates[:, a0:a1], w_flat[a0:a1], m=T, out=tmp)
                ops.linear_dinput(dgates[:, b0:b1], w_flat[b0:b1], m=T, out=tmp, accumulate=True)
                unit = torch.arange(T + 1, dtype=torch.int64, device=x.device)
                for j in range(n_in):     # dx[slots[j, t]] += d x_flat[t, slot j]
                    ops.gather_reduce_bwd(tmp[:, j * ld:j * ld + fdim], unit, slots[j], T, "sum", dx)
        return dx, None, dw, dbias


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
    if node_map is not None and x_src.requires_grad:
        raise RuntimeError("gradient w.r.t. a node_map'ed feature table is not supported")
    slots, nonempty = ops.segment_slots(plan.seg_ptr, plan.src, num_targets, n_in, node_map=node_map)
    bias = layer.lstm.bias_ih + layer.lstm.bias_hh
    flat = (node_map is None and fdim % 16 == 0 and x_src.is_contiguous()
            and 4 * num_targets * n_in * fdim <= _FLAT_BYTES)
    gates = (_LstmGatesFlatFn if flat else _LstmGatesFn).apply(x_src, slots, layer.lstm.weight_ih, bias)
    return _LstmCell0Fn.apply(gates, nonempty, fdim)
