"""Static-shape training engine for the sampled mini-batch step (the fast path the benchmark
times).  One step = fig2a's definition (reference
experiments/fig2a/runners/graphsage_runners.py:38-46): sample -> gather -> forward -> cross
entropy on the seed rows -> backward -> Adam, supervised term only.

It computes exactly what GraphSAGE.forward_batch + F.cross_entropy + autograd + Adam compute
(tests/test_gpu_engine.py checks that), but:
  * nothing is sized by the batch: every buffer has the capacity the fan-outs imply and the
    kernels read the batch's node/edge counts from the sampler's device state block, so the
    host never synchronises and the step can be replayed from a HIP graph;
  * layer i is evaluated only on the hop_nodes[L-1-i] rows that can reach the seeds;
  * layer 0 reads neighbour and root rows straight from the global feature table through the
    batch's node vector — batch.x is never materialised;
  * all gradients live in one flat buffer (one all-reduce under data parallelism, one Adam
    launch).
"""
import torch

from . import ops

_REDUCE = {'mean': 'mean', 'sum': 'sum'}


def flatten_parameters(model):
    """Move every parameter of `model` into one flat fp32 buffer (views keep their shapes)."""
    params = [p for p in model.parameters()]
    total = sum(p.numel() for p in params)
    flat = torch.empty(total, dtype=torch.float32, device=params[0].device)
    off = 0
    for p in params:
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat[off:off + n].view(p.shape)
        off += n
    return flat


class SageTrainEngine:
    def __init__(self, model, sampler, features, labels, batch_size, lr=1e-3, betas=(0.9, 0.999),
                 eps=1e-8, weight_decay=0.0, world_size=1, process_group=None):
        if model.aggregator not in _REDUCE:
            raise NotImplementedError(f"engine supports {sorted(_REDUCE)}, got {model.aggregator}")
        self.model, self.sampler = model, sampler
        self.reduce = _REDUCE[model.aggregator]
        self.features, self.labels = features, labels
        self.B = int(batch_size)
        self.L = len(model.layers)
        fan = sampler.num_neighbors
        if len(fan) != self.L or any(k < 0 for k in fan):
            raise NotImplementedError("engine needs one non-negative fan-out per layer")
        dev = features.device
        self.dev = dev
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.world_size, self.pg = world_size, process_group
        # capacities: hop_cap[h] = max nodes after hop h
        self.hop_cap = [self.B]
        frontier = self.B
        for k in fan:
            frontier = frontier * k
            self.hop_cap.append(self.hop_cap[-1] + frontier)
        self.buf = sampler.buffers(self.B)
        st = self.buf.state
        self.n_dev = [st[ops.ST_HOP_NODES + h: ops.ST_HOP_NODES + h + 1] for h in range(self.L + 1)]

        self.flat = flatten_parameters(model)
        self.grad = torch.zeros_like(self.flat)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        # gradient views with the parameters' shapes
        self.gviews = {}
        off = 0
        for p in model.parameters():
            self.gviews[p] = self.grad[off:off + p.numel()].view(p.shape)
            off += p.numel()

        def padded(rows, cols):
            ld = (cols + 15) // 16 * 16
            return torch.zeros((rows, ld), dtype=torch.float32, device=dev)[:, :cols]

        self.act = []   # per layer: dict of static buffers
        for i, layer in enumerate(model.layers):
            t_cap = self.hop_cap[self.L - 1 - i]
            fin = layer.lin_l.weight.size(1)
            fout = layer.lin_l.weight.size(0)
            a = dict(t_cap=t_cap, fin=fin, fout=fout, t_dev=self.n_dev[self.L - 1 - i],
                     agg=padded(t_cap, fin), z=padded(t_cap, fout), y=padded(t_cap, fout),
                     inv=torch.zeros(t_cap, dtype=torch.float32, device=dev),
                     dz=padded(t_cap, fout))
            if i < self.L - 1:
                a['h'] = padded(t_cap, fout)
                a['dh'] = padded(t_cap, fout)
            if i > 0:
                a['dagg'] = padded(t_cap, fin)
            self.act.append(a)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.seeds = torch.zeros(self.B, dtype=torch.int64, device=dev)
        self.counter = 0

    # ------------------------------------------------------------------ pieces
    def sample(self, seeds, counter):
        self.seeds.copy_(seeds)
        self.sampler.sample_async(self.seeds, counter=counter)

    def forward_backward(self):
        L, buf = self.L, self.buf
        x, node_map = self.features, buf.node
        lib = ops
        for i, layer in enumerate(self.model.layers):
            a = self.act[i]
            t_cap, t_dev = a['t_cap'], a['t_dev']
            lib.gather_reduce_fwd(x, buf.seg_ptr, buf.src, t_cap, self.reduce, node_map=node_map,
                                  num_targets_dev=t_dev, out=a['agg'], want_argmax=False)
            lib.linear_concat_fwd(a['agg'], layer.lin_l.weight, x, layer.lin_r.weight, m=t_cap,
                                  m_dev=t_dev, rows2=node_map, out=a['z'])
            relu = i < L - 1
            lib.l2norm_relu_fwd(a['z'], relu, m=t_cap, m_dev=t_dev, y=a['y'], h=a.get('h'), inv_norm=a['inv'])
            x, node_map = a.get('h'), None
        last = self.act[-1]
        lib.softmax_xent(last['y'], self.labels, label_rows=buf.node, m=self.B, loss=self.loss,
                         dlogits=last['dz'], checked=True)
        # backward; `g` = gradient w.r.t. the layer's (normalised, ReLU'd) output
        self.grad.zero_()
        g = last['dz']
        for i in range(L - 1, -1, -1):
            a, layer = self.act[i], self.model.layers[i]
            t_cap, t_dev = a['t_cap'], a['t_dev']
            relu = i < L - 1
            lib.l2norm_relu_bwd(g, a['y'], a['inv'], relu, m=t_cap, m_dev=t_dev, dz=a['dz'])
            x_in = self.features if i == 0 else self.act[i - 1]['h']
            rows = buf.node if i == 0 else None
            lib.linear_dweight(a['dz'], a['agg'], m=t_cap, m_dev=t_dev, dw=self.gviews[layer.lin_l.weight])
            lib.linear_dweight(a['dz'], x_in, m=t_cap, m_dev=t_dev, rows=rows, dw=self.gviews[layer.lin_r.weight])
            if i > 0:
                prev = self.act[i - 1]
                prev['dh'].zero_()
                lib.linear_dinput(a['dz'], layer.lin_r.weight, m=t_cap, m_dev=t_dev, out=prev['dh'])
                lib.linear_dinput(a['dz'], layer.lin_l.weight, m=t_cap, m_dev=t_dev, out=a['dagg'])
                lib.gather_reduce_bwd(a['dagg'], buf.seg_ptr, buf.src, t_cap, self.reduce, prev['dh'],
                                      num_targets_dev=t_dev)
                g = prev['dh']

    def all_reduce(self):
        if self.world_size > 1:
            torch.distributed.all_reduce(self.grad, group=self.pg)

    def optimizer_step(self):
        ops.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.step_dev, self.lr,
                      self.betas[0], self.betas[1], self.eps, self.weight_decay,
                      grad_scale=1.0 / self.world_size)

    def step(self, seeds, counter=None):
        """One full training step on `seeds` (device int64[B]); returns the device loss tensor."""
        if counter is None:
            counter = self.counter
            self.counter += 1
        self.sample(seeds, counter)
        self.forward_backward()
        self.all_reduce()
        self.optimizer_step()
        return self.loss
