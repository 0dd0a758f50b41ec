"""SAGE layer / GraphSAGE model / batch assembly / training step restated on CPU torch
(fp32, autograd), unfused and as written in the reference:

  filter_data          -> batch_assemble      reference samplers/uniform_loader.py:235-249
  SAGE.message         -> x_j = x[ei[0]]      reference layers/sage.py:136-137
  SAGE.aggregate       -> scatter(...)        reference layers/sage.py:195-210
                          [upstream torch_scatter.scatter: out-of-place, dim_size rows,
                           rows that receive nothing are 0 for sum/mean/max;
                           mean = sum / clamp(count, min=1)]
  SAGE.forward         -> sage_forward        reference layers/sage.py:99-134
  GraphSAGE.forward    -> graphsage_forward   reference models/graphsage.py:44-49
  fig2a train step     -> train_step          reference experiments/fig2a/runners/graphsage_runners.py:38-46
  unsup loss           -> unsup_loss          reference trainers/base_trainers.py:170-181

Test infrastructure only (see oracle/__init__.py).
"""
import torch
import torch.nn.functional as F


def batch_assemble(x, y, node, src_local, dst_local):
    """filter_data: node-level attributes indexed by `node`; edge_index = [src; dst]."""
    node = torch.as_tensor(node, dtype=torch.long)
    xb = x.index_select(0, node)
    yb = y.index_select(0, node) if y is not None else None
    edge_index = torch.stack([torch.as_tensor(src_local, dtype=torch.long),
                              torch.as_tensor(dst_local, dtype=torch.long)], dim=0)
    return xb, yb, edge_index


def scatter_ref(x_j, index, dim_size, reduce):
    """torch_scatter.scatter(x_j, index, dim=0, dim_size=dim_size, reduce=reduce)."""
    Fdim = x_j.size(1)
    if reduce in ("sum", "add"):
        return torch.zeros(dim_size, Fdim, dtype=x_j.dtype).index_add(0, index, x_j)
    if reduce == "mean":
        s = torch.zeros(dim_size, Fdim, dtype=x_j.dtype).index_add(0, index, x_j)
        cnt = torch.zeros(dim_size, dtype=x_j.dtype).index_add(0, index, torch.ones_like(index, dtype=x_j.dtype))
        return s / cnt.clamp(min=1).unsqueeze(1)
    if reduce == "max":
        out = torch.zeros(dim_size, Fdim, dtype=x_j.dtype)
        if index.numel() == 0:
            return out
        # include_self=False: rows that receive values ignore the zero init; untouched rows stay 0
        return out.scatter_reduce(0, index.unsqueeze(1).expand(-1, Fdim), x_j, "amax", include_self=False)
    raise ValueError(reduce)


def sage_forward(x, edge_index, lin_l, lin_r=None, aggregator="mean", pool_w=None, pool_b=None,
                 normalize=True, lstm=None):
    """One SAGE layer on all rows of x.  lin_l/lin_r: [out, in] weights, no bias
    (reference layers/sage.py:78-81).  gcn: self-loops for unique targets, no root term
    (:62-65, :114-117).  *_pool: relu(pool(x_j)) per edge before the reduce (:201-207)."""
    n = x.size(0)
    ei = edge_index
    if aggregator == "gcn":
        roots = torch.unique(ei[1])
        ei = torch.cat([ei, torch.stack([roots, roots], dim=0)], dim=1)
    if aggregator == "lstm":
        # lstm = dict(n_in, w_ih, w_hh, b_ih, b_hh); rows without in-edges are 0 (:121-122)
        out = lstm_aggregate_ref(x, ei, n, lstm["n_in"], lstm["w_ih"], lstm["w_hh"], lstm["b_ih"], lstm["b_hh"])
        out = F.linear(out, lin_l) + F.linear(x, lin_r)
        return F.normalize(out, p=2.0, dim=-1) if normalize else out
    x_j = x.index_select(0, ei[0])
    if aggregator in ("mean", "max", "sum"):
        reduce = aggregator
    elif aggregator == "gcn":
        reduce = "mean"
    elif aggregator == "max_pool":
        x_j = F.relu(F.linear(x_j, pool_w, pool_b))
        reduce = "max"
    elif aggregator == "mean_pool":
        x_j = F.relu(F.linear(x_j, pool_w, pool_b))
        reduce = "mean"
    else:
        raise ValueError(aggregator)
    out = scatter_ref(x_j, ei[1], n, reduce)
    out = F.linear(out, lin_l)
    if aggregator != "gcn" and lin_r is not None:
        out = out + F.linear(x, lin_r)
    if normalize:
        out = F.normalize(out, p=2.0, dim=-1)
    return out


def graphsage_forward(x, edge_index, layers, aggregator="mean"):
    """layers: list of dicts with keys lin_l, lin_r (absent for gcn), pool_w, pool_b.
    Same edge_index for every layer, ReLU between layers, every layer normalised."""
    for i, p in enumerate(layers):
        x = sage_forward(x, edge_index, p["lin_l"], p.get("lin_r"), aggregator,
                         p.get("pool_w"), p.get("pool_b"))
        if i < len(layers) - 1:
            x = x.relu()
    return x


def train_step_loss(x_table, y_table, node, src_local, dst_local, batch_size, layers,
                    aggregator="mean"):
    """fig2a's step up to the loss: assemble, forward over ALL sampled rows, slice [:B], CE."""
    xb, yb, ei = batch_assemble(x_table, y_table, node, src_local, dst_local)
    y_hat = graphsage_forward(xb, ei, layers, aggregator)[:batch_size]
    return F.cross_entropy(y_hat, yb[:batch_size]), y_hat


def adam_ref(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8,
             weight_decay=0.0):
    """torch.optim.Adam single-tensor update (no amsgrad), step is 1-based."""
    if weight_decay != 0.0:
        grad = grad + weight_decay * param
    exp_avg = beta1 * exp_avg + (1 - beta1) * grad
    exp_avg_sq = beta2 * exp_avg_sq + (1 - beta2) * grad * grad
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = exp_avg_sq.sqrt() / (bc2 ** 0.5) + eps
    param = param - (lr / bc1) * (exp_avg / denom)
    return param, exp_avg, exp_avg_sq


def unsup_loss(out, pos_out, neg_out):
    """reference trainers/base_trainers.py:179-181."""
    pos_loss = F.logsigmoid((out * pos_out).sum(-1)).mean()
    neg_loss = F.logsigmoid(-(out * neg_out).sum(-1)).mean()
    return -pos_loss - neg_loss


def lstm_aggregate_ref(x, edge_index, num_nodes, n_in, w_ih, w_hh, b_ih, b_hh):
    """The computation reference layers/sage.py:147-193 is after, made well-defined (per target,
    neighbours in edge order grouped by target, first n_in kept, padded with the first row,
    flattened, ONE LSTMCell step from a zero state; no in-edge -> 0).  Parity unpinned: the
    reference's own code is disabled upstream as broken (settings.py:48-50)."""
    Fdim = x.size(1)
    order = torch.argsort(edge_index[1], stable=True)
    src, dst = edge_index[0][order], edge_index[1][order]
    out = torch.zeros(num_nodes, Fdim, dtype=x.dtype)
    rows_out = []
    for t in range(num_nodes):
        nb = src[dst == t][:n_in]
        if nb.numel() == 0:
            rows_out.append(torch.zeros(Fdim, dtype=x.dtype))
            continue
        rows = x[nb]
        if rows.size(0) < n_in:
            rows = torch.cat([rows, rows[0:1].repeat(n_in - rows.size(0), 1)])
        flat = rows.reshape(1, -1)
        gates = F.linear(flat, w_ih, b_ih) + F.linear(torch.zeros(1, Fdim, dtype=x.dtype), w_hh, b_hh)
        i, f, g, o = gates.chunk(4, dim=1)
        c = torch.sigmoid(i) * torch.tanh(g)
        rows_out.append((torch.sigmoid(o) * torch.tanh(c)).squeeze(0))
    return torch.stack(rows_out) if rows_out else out
