"""autograd Functions over the HIP kernels, and the EdgePlan the layers share.

These are what SAGE.forward composes (reference layers/sage.py:99-134): gather+reduce,
self||neighbour Linear, L2 normalise (+ReLU).  Backward formulas are the ones autograd
derives for the reference's unfused ops; tests compare both against the CPU oracle.
"""
import torch
from torch.autograd import Function

from . import ops


class EdgePlan:
    """Target-CSR view of an edge_index plus, for sampled batches, the hop structure.

    seg_ptr[t]..seg_ptr[t+1] are the (target-sorted) edges into local node t; src holds their
    source ids in that order.  hop_nodes[h] = number of nodes after hop h (hop 0 = seeds),
    hop_edges[h] likewise.  node/table let the first layer gather straight from the global
    feature table (node_map path) instead of a materialised batch.x.
    """

    def __init__(self, seg_ptr, src, num_dst, num_src, perm=None, hop_nodes=None, hop_edges=None,
                 node=None, table=None):
        self.seg_ptr, self.src, self.perm = seg_ptr, src, perm
        self.num_dst, self.num_src = int(num_dst), int(num_src)
        self.hop_nodes, self.hop_edges = hop_nodes, hop_edges
        self.node, self.table = node, table
        self.key = None


def plan_for(edge_index, num_src, num_dst):
    """Plan attached by the loader, or built (and cached on the tensor) for any edge_index."""
    plan = getattr(edge_index, "_sage_plan", None)
    if not isinstance(edge_index, torch.Tensor) or edge_index.layout != torch.strided:
        if plan is None:
            plan = _plan_from_adj_t(edge_index, num_src, num_dst)
            try:
                edge_index._sage_plan = plan
            except Exception:
                pass
        if plan.num_dst < num_dst:
            raise RuntimeError(f"adj_t has {plan.num_dst} rows, the layer needs {num_dst} targets")
        return plan
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape))
    if plan is not None and (plan.key is None or plan.key == key) and plan.num_dst >= num_dst:
        return plan
    if edge_index.dim() != 2 or edge_index.size(0) != 2 or edge_index.dtype != torch.int64:
        raise RuntimeError("edge_index must be an int64 tensor of shape [2, E]")
    if not edge_index.is_cuda:
        raise RuntimeError("graphsage_amd runs on the GPU only: move edge_index to the device")
    ei = edge_index.contiguous()
    if ei.size(1) > 0:
        lo, hi = int(ei[0].min()), int(ei[0].max())
        if lo < 0 or hi >= num_src:
            raise RuntimeError(f"edge_index[0] out of range [0,{num_src})")
    seg_ptr, src, perm = ops.coo_to_csr(ei[0], ei[1], num_dst)
    plan = EdgePlan(seg_ptr, src, num_dst, num_src, perm=perm)
    plan.key = key
    try:
        edge_index._sage_plan = plan
    except Exception:  # tensors that refuse attributes: just do not cache
        pass
    return plan


def _plan_from_adj_t(adj_t, num_src, num_dst):
    """The SparseTensor form of the layer's input (reference layers/sage.py:107-109, :213-232:
    `matmul(adj_t, x, reduce)` with adj_t = transposed adjacency, row = target, col = source).
    Its CSR arrays ARE the target-grouped plan, so nothing is sorted.  Accepts a
    torch.sparse_csr tensor or any object with torch_sparse's `.csr()` -> (rowptr, col, value);
    values are ignored as the reference drops them (:218)."""
    if isinstance(adj_t, torch.Tensor):
        if adj_t.layout != torch.sparse_csr:
            raise RuntimeError("sparse edge_index must be a torch.sparse_csr tensor (adj_t: row = target)")
        rowptr, col = adj_t.crow_indices(), adj_t.col_indices()
        rows, cols = adj_t.shape[-2], adj_t.shape[-1]
    elif hasattr(adj_t, "csr"):
        rowptr, col = adj_t.csr()[:2]
        rows, cols = rowptr.numel() - 1, num_src
        if hasattr(adj_t, "sparse_sizes"):
            rows, cols = adj_t.sparse_sizes()
    else:
        raise RuntimeError("edge_index must be an int64 tensor of shape [2, E] or a CSR adjacency (adj_t)")
    if not rowptr.is_cuda:
        raise RuntimeError("graphsage_amd runs on the GPU only: move adj_t to the device")
    rowptr, col = rowptr.to(torch.int64).contiguous(), col.to(torch.int64).contiguous()
    if rowptr.dim() != 1 or rowptr.numel() != rows + 1 or cols > num_src:
        raise RuntimeError(f"adj_t of shape [{rows}, {cols}] does not fit {num_src} source rows")
    if col.numel() > 0:
        lo, hi, first, last = int(col.min()), int(col.max()), int(rowptr[0]), int(rowptr[-1])
        if lo < 0 or hi >= num_src or first != 0 or last != col.numel() or bool((rowptr[1:] < rowptr[:-1]).any()):
            raise RuntimeError("adj_t: column ids out of range or rowptr not a scan of the row lengths")
    return EdgePlan(rowptr, col, rows, num_src)


class GatherReduceFn(Function):
    @staticmethod
    def forward(ctx, x, seg_ptr, src, num_targets, reduce, node_map):
        x = x if x.stride(-1) == 1 else x.contiguous()
        out, argmax = ops.gather_reduce_fwd(x, seg_ptr, src, num_targets, reduce, node_map=node_map)
        ctx.save_for_backward(seg_ptr, src, argmax if argmax is not None else seg_ptr)
        ctx.meta = (int(num_targets), reduce, tuple(x.shape), node_map is not None, argmax is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        seg_ptr, src, argmax = ctx.saved_tensors
        num_targets, reduce, xshape, mapped, has_arg = ctx.meta
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None, None
        if mapped:
            raise RuntimeError("gradient w.r.t. a node_map'ed feature table is not supported")
        dx = ops.empty_rows(xshape[0], xshape[1], dout.device, zero=True)
        ops.gather_reduce_bwd(ops.aligned_rows(dout), seg_ptr, src, num_targets, reduce, dx,
                              argmax=argmax if has_arg else None)
        return dx, None, None, None, None, None


class LinearConcatFn(Function):
    """out = a1[rows1] @ w1^T + a2[rows2] @ w2^T (+ bias) (relu)."""

    @staticmethod
    def forward(ctx, a1, w1, a2, w2, bias, rows1, rows2, relu, m):
        # rows that cannot be read 16 bytes at a time (F = 602 with stride 602) cost the GEMM ~1.7x:
        # re-lay weights and un-mapped operands with padded rows first (a mapped operand is a whole
        # feature table: the loader already keeps that one padded)
        a1 = ops.aligned_rows(a1) if rows1 is None else (a1 if a1.stride(-1) == 1 else a1.contiguous())
        if a2 is not None:
            a2 = ops.aligned_rows(a2) if rows2 is None else (a2 if a2.stride(-1) == 1 else a2.contiguous())
        w1 = ops.aligned_rows(w1)
        w2 = ops.aligned_rows(w2) if w2 is not None else None
        out = ops.linear_concat_fwd(a1, w1, a2, w2, bias=bias, relu=relu, m=m, rows1=rows1, rows2=rows2)
        ctx.save_for_backward(a1, w1, a2 if a2 is not None else a1, w2 if w2 is not None else w1,
                              rows1 if rows1 is not None else w1, rows2 if rows2 is not None else w1,
                              out if relu else w1)
        ctx.meta = (a2 is not None, rows1 is not None, rows2 is not None, bool(relu), bias is not None,
                    int(out.size(0)))
        return out

    @staticmethod
    def backward(ctx, dout):
        a1, w1, a2, w2, rows1, rows2, out = ctx.saved_tensors
        has2, has_r1, has_r2, relu, has_bias, m = ctx.meta
        g = dout
        if relu:
            g = torch.ops.aten.threshold_backward(dout, out, 0.0)      # dout * [out > 0], one pass
        g = ops.aligned_rows(g)
        need = ctx.needs_input_grad
        da1 = dw1 = da2 = dw2 = dbias = None
        if has_bias and need[4]:
            dbias = torch.zeros(w1.size(0), dtype=torch.float32, device=g.device)
        if need[0]:
            if has_r1:
                raise RuntimeError("gradient w.r.t. a row-mapped operand is not supported")
            da1 = torch.zeros_like(a1) if a1.size(0) != m else None
            if da1 is None:
                da1 = ops.linear_dinput(g, w1, m=m)
            else:
                ops.linear_dinput(g, w1, m=m, out=da1[:m])
        if need[1]:
            dw1 = ops.linear_dweight(g, a1, m=m, rows=rows1 if has_r1 else None, dbias=dbias)
            dbias_done = True
        else:
            dbias_done = False
        if has2:
            if need[2]:
                if has_r2:
                    raise RuntimeError("gradient w.r.t. a row-mapped operand is not supported")
                if a2.size(0) != m:
                    da2 = torch.zeros_like(a2)
                    ops.linear_dinput(g, w2, m=m, out=da2[:m])
                else:
                    da2 = ops.linear_dinput(g, w2, m=m)
            if need[3]:
                dw2 = ops.linear_dweight(g, a2, m=m, rows=rows2 if has_r2 else None)
        if dbias is not None and not dbias_done:
            dbias = g[:m].sum(0)
        return da1, dw1, da2, dw2, dbias, None, None, None, None


class L2NormReLUFn(Function):
    """normalize(z) (and ReLU when relu=True) — reference layers/sage.py:131-132 and the
    x.relu_() of models/graphsage.py:47-48."""

    @staticmethod
    def forward(ctx, z, relu):
        z = z.contiguous()
        y, h, inv = ops.l2norm_relu_fwd(z, relu)
        ctx.save_for_backward(y, inv)
        ctx.relu = bool(relu)
        return h if relu else y

    @staticmethod
    def backward(ctx, dh):
        y, inv = ctx.saved_tensors
        return ops.l2norm_relu_bwd(dh.contiguous(), y, inv, ctx.relu), None


def gather_reduce(x, plan_or_ptr, src=None, num_targets=None, reduce="mean", node_map=None):
    if isinstance(plan_or_ptr, EdgePlan):
        p = plan_or_ptr
        return GatherReduceFn.apply(x, p.seg_ptr, p.src, p.num_dst if num_targets is None else num_targets,
                                    reduce, node_map)
    return GatherReduceFn.apply(x, plan_or_ptr, src, num_targets, reduce, node_map)


def linear_concat(a1, w1, a2=None, w2=None, bias=None, rows1=None, rows2=None, relu=False, m=None):
    return LinearConcatFn.apply(a1, w1, a2, w2, bias, rows1, rows2, relu, m)


def l2norm_relu(z, relu=False):
    return L2NormReLUFn.apply(z, relu)
