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
    launch);
  * the class-logit layer of the supervised mean / sum step is ONE launch
    (sage_logit_layer_fwd_bwd: reduce, both weight products, normalise, cross entropy and, in
    the same wave, the seed's input gradient — rows of the hidden gradient that only one seed
    writes are stored, not added, sage_count_row_writers), its weight gradients and the loss
    mean ride in front of the next launch's grid (sage_l2norm_relu_bwd_z_dw), and the hidden
    layer below it is never normalised as a whole: the forward kernel normalises the rows it
    gathers, the backward works from the pre-activation rows;
  * on one GPU the layer-0 weight gradients never pass through atomics: the split-K slices are
    stored and Adam sums them as it reads them (sage_linear_dweight2_parts /
    sage_adam_step_parts); the Adam launch carries the end-of-step event as its completion
    signal (sage_adam_step_signal);
  * the pooling aggregators keep P = relu(pool(x)) per SOURCE row of a layer in a static buffer
    and route d agg back through the reduce and the ReLU in one edge pass
    (sage_pool_linear_fwd / sage_pool_max_bwd / sage_pool_mean_bwd); their large weight
    gradients run on a forked stream beside that pass; at Reddit size layer 0's edge pass is
    pulled by the source rows from per-source edge lists built behind the sampler
    (sage_pool_reverse_edges / sage_pool_*_bwd_pull), and their last layer is one launch like the
    class-logit layer above (sage_logit_layer_pool_fwd_bwd); evaluate() computes relu(pool(x)) once
    over all nodes when the batches together exceed the graph;
  * the lstm aggregator (reference layers/sage.py:67-69, :147-193 as read in layers/lstm.py)
    materialises the cell's flattened input x_flat [targets, n_in * F] once per layer and step
    (sage_lstm_gather_flat: 0.8 GB at the Reddit shape, static, in 288 GB of HBM), so that the
    gates contraction and its weight gradient are ONE launch per gate block over K = n_in * F with
    both operands contiguous in k — W_ih lives in the flat parameter buffer with padded rows and
    is never re-laid, its gradient goes straight into the flat gradient, Adam is the fused kernel;
  * the trainers' three-forward steps (unsup=True: supervised + unsupervised term, reference
    trainers/node_level_graphsage_trainers.py:34-46; unsup='only': :104-117) read the positive /
    negative variants' layer-0 rows through node maps (no second and third batch.x), run those two
    variants on streams of their own beside the batch's, forward and backward, and take the serial
    schedule (overlap_gather='auto', gather_pays_overlap()).
"""
import contextlib
import ctypes
import os
import time

import torch

from . import _lib, ops
from .dist import PendingReduce, all_reduce_flat

_SPIN_CYCLES = 1500000


def _runs_beside(a, b):
    """Do kernels on streams a and b run concurrently?  HIP deals its streams onto a few hardware
    queues (GPU_MAX_HW_QUEUES, 4 by default), and two streams of one queue take turns: measured
    with a spin kernel on each, two take the time of one on different queues and of two on a
    shared one (scratch/stream_queues.py: every fourth stream of torch's pool shares the default
    stream's queue)."""
    def timed(streams):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for st in streams:
            with torch.cuda.stream(st):
                torch.cuda._sleep(_SPIN_CYCLES)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    timed([a])
    one = min(timed([a]), timed([a]))
    two = min(timed([a, b]), timed([a, b]))
    return two < 1.7 * one


def new_side_stream(dev, beside):
    """A stream whose kernels can run beside those of the streams in `beside`.  An engine whose
    gather stream shares the main stream's hardware queue loses the overlap AND stalls on its own
    cross-stream events: 0.32 ms per step against 0.25 (and 0.26 on one stream), which is what
    every fourth engine of a process got before this check (scratch/placement.py).
    SAGE_NO_STREAM_PROBE=1 takes the next stream of torch's pool unseen."""
    st = torch.cuda.Stream(device=dev)
    if os.environ.get('SAGE_NO_STREAM_PROBE') or not hasattr(torch.cuda, '_sleep'):
        return st
    for _ in range(8):
        if all(_runs_beside(o, st) for o in beside):
            return st
        st = torch.cuda.Stream(device=dev)
    return st

# gcn: no root weight (lin_r); the pooling aggregators reduce relu(pool(x_j)) instead of x_j
# lstm: no segment reduce at all — the aggregate is one LSTMCell step over the flattened neighbour rows
_REDUCE = {'mean': 'mean', 'sum': 'sum', 'max': 'max', 'gcn': 'gcn', 'max_pool': 'max', 'mean_pool': 'mean',
           'lstm': None}


def flatten_parameters(model, pad_to=16):
    """Move every parameter of `model` into one flat fp32 buffer.  Matrices are stored with
    their row stride rounded up to `pad_to` floats (602 -> 608) so that every row is 64-byte
    aligned and the GEMM kernels can use 16-byte loads; the parameters become [out, in] views
    of that storage (same values, same state_dict).  The padding stays zero under Adam."""
    params = [p for p in model.parameters()]
    dev = params[0].device

    def ld_of(p):
        return (p.size(1) + pad_to - 1) // pad_to * pad_to if p.dim() == 2 else None

    def vec_len(p):      # vectors (the pool bias) padded as well: what follows stays 64-byte aligned
        return (p.numel() + pad_to - 1) // pad_to * pad_to

    total = sum(p.size(0) * ld_of(p) if p.dim() == 2 else vec_len(p) for p in params)
    flat = torch.zeros(total, dtype=torch.float32, device=dev)
    layout = []
    off = 0
    for p in params:
        if p.dim() == 2:
            ld = ld_of(p)
            n = p.size(0) * ld
            view = flat[off:off + n].view(p.size(0), ld)[:, :p.size(1)]
        else:
            n = vec_len(p)
            view = flat[off:off + p.numel()].view(p.shape)
        view.copy_(p.data)
        p.data = view
        layout.append((off, n, tuple(p.shape), ld_of(p)))
        off += n
    return flat, layout


def views_like(flat, layout):
    out = []
    for off, n, shape, ld in layout:
        if ld is not None:
            out.append(flat[off:off + n].view(shape[0], ld)[:, :shape[1]])
        else:
            numel = 1
            for d in shape:
                numel *= d
            out.append(flat[off:off + numel].view(shape))
    return out


class SageTrainEngine:
    """See the module docstring.  Two execution modes, same arithmetic:

    * `step(seeds, counter)` — everything on the current stream, one batch at a time;
    * pipelined — the sampler of batch i+1 runs on a side stream into the other of two
      sample-buffer slots while batch i computes (sampling needs no weights, and its ~20 short
      latency-bound kernels hide completely behind the MFMA-bound GEMMs).  Driven eagerly with
      `prefetch()` + `pipelined_step()`, or replayed from HIP graphs with `capture()` +
      `replay()`.
    """

    @staticmethod
    def gather_pays_overlap(num_features, batch_size, fanouts, unsup=False):
        """The overlap_gather='auto' rule.  The side-stream gather costs a cross-stream hand-shake per
        step and pays when it hides more than that — measured, same box, ms per step overlapped /
        serial (profiles/r3_schedule_ab.txt):
          supervised step  Reddit rows (602 floats, 351 MB of gather capacity per batch) 0.213 / 0.225,
                           products rows (100 floats, 58 MB) 0.097 / 0.103, PPI rows (50 floats, 29 MB) 0.187 / 0.158
          three forwards   (unsup=True / 'only': one of the three layer-0 gathers can go aside)
                           Reddit 0.855 / 0.831, products 0.500 / 0.477
        hence: the supervised step with at least 40 MB of capacity (4 bytes x feature width x sampled edges)."""
        if unsup:
            return False
        frontier, edge_cap = int(batch_size), 0
        for k in fanouts:
            frontier *= int(k)
            edge_cap += frontier
        return 4 * int(num_features) * edge_cap >= (40 << 20)

    def __init__(self, model, sampler, features, labels, batch_size, lr=1e-3, betas=(0.9, 0.999),
                 eps=1e-8, weight_decay=0.0, world_size=1, process_group=None, overlap_gather=False,
                 keep_grads=False, group=4, unsup=False, force_collective=False):
        if model.aggregator not in _REDUCE:
            raise NotImplementedError(f"engine supports {sorted(_REDUCE)}, got {model.aggregator}")
        self.model = model
        self.reduce = _REDUCE[model.aggregator]
        self.root = model.aggregator != 'gcn'      # gcn layers have no lin_r (reference layers/sage.py:62-65)
        # max_pool / mean_pool (reference layers/sage.py:75-76, :201-207): relu(pool(x_j)) is
        # row-wise, so the pool Linear runs once per SOURCE NODE of the layer (its own static
        # buffer P), the segments then reduce rows of P; backward: scatter through the reduce
        # into dP, relu mask, pool weight / bias gradients, and (above layer 0) the input gradient
        self.pool = model.aggregator in ('max_pool', 'mean_pool')
        if self.pool and unsup:
            raise NotImplementedError("engine: the pooling aggregators run the supervised (fig2a) step only")
        self.lstm = model.aggregator == 'lstm'
        if self.lstm and unsup:
            raise NotImplementedError("engine: the lstm aggregator runs the supervised (fig2a) step only")
        if overlap_gather not in (True, False, 'auto'):
            raise ValueError("overlap_gather: True, False or 'auto'")
        if self.pool or self.lstm:
            overlap_gather = False    # the layer-0 aggregate needs this step's weights (pool Linear / W_ih)
        # Data parallel: the gradient all-reduce of step i is launched behind the backward's last
        # kernel and its Adam step sits in front of the first kernel of step i+1 that reads the
        # weights.  The one piece of weight-free work that can run on the main stream in between
        # is the layer-0 gather of batch i+1 (53 us of HBM traffic at the Reddit shape against a
        # latency-bound 1.3 MB collective): with a collective the engine therefore runs the
        # SERIAL schedule — all-reduce launch -> gather -> wait -> Adam -> forward contraction —
        # and not overlap_gather, whose gather of batch i+1 has already run beside step i's
        # contraction and leaves the collective fully exposed.  force_collective: take the
        # collective branch with one rank as well (tests: a 1-rank RCCL group on the one-GPU box).
        self.collective = world_size > 1 or bool(force_collective)
        if self.collective:
            overlap_gather = False
        # The class-logit layer (the last SAGE layer on the B seed rows) in one launch each way
        # (sage_logit_layer_fwd / _bwd) instead of gather / split-K contraction / fused loss and
        # d-input pair / scatter-add / loss mean: supervised step, mean or sum, two or more layers
        last_w = model.layers[-1].lin_l.weight
        # multi-label targets (PPI: fp32 [N, C] of 0 / 1, BCEWithLogitsLoss on the normalised logits —
        # reference examples/ppi_graphsage_supervised.py:30) instead of int64 class ids + cross entropy
        self.multilabel = labels is not None and labels.dim() == 2
        if self.multilabel:
            if labels.dtype != torch.float32 or labels.size(1) != last_w.size(0):
                raise RuntimeError("engine: multi-label targets must be fp32 [N, out_channels]")
            if unsup is True:
                raise NotImplementedError("engine: the trainers' combined loss is implemented for single-label targets")
        # SAGE_NO_FUSED_LOGIT=1 keeps the separate kernels (the A/B switch of scratch/hb.sh)
        # (max: only in the one-launch forward + backward form — the arg-max of every column lives in
        # the forward's registers — so not with any of the switches that take that form apart)
        max_ok = not any(os.environ.get(k) for k in ('SAGE_NO_FUSED_LOGIT_DW', 'SAGE_NO_PRENORM', 'SAGE_NO_LOGIT_FWDBWD',
                                                     'SAGE_NO_FUSED_LOGIT_MAX'))
        self.fused_logit = (not os.environ.get('SAGE_NO_FUSED_LOGIT') and not unsup and not self.pool
                            and not self.multilabel
                            and ((self.root and self.reduce in ('mean', 'sum')) or (self.reduce in ('max', 'gcn') and max_ok))
                            and len(model.layers) >= 2
                            and ops.logit_layer_supported(last_w.size(1), last_w.size(0)))
        # ... with the layer's weight gradients in the backward's launch (SAGE_NO_FUSED_LOGIT_DW=1: a launch of their own)
        self.fused_logit_dw = self.fused_logit and not os.environ.get('SAGE_NO_FUSED_LOGIT_DW')
        # ... and reads the hidden layer's PRE-activation rows (SAGE_NO_PRENORM=1: normalise pass kept)
        self.prenorm = self.fused_logit and not os.environ.get('SAGE_NO_PRENORM')
        # ... and, with those, needs no backward launch of its own at all: every seed's wave runs its
        # backward behind its forward (sage_logit_layer_fwd_bwd), the layer's weight gradients and the
        # loss mean ride in front of the hidden layer's normalise-backward pass
        # (sage_l2norm_relu_bwd_z_dw).  SAGE_NO_LOGIT_FWDBWD=1: forward and backward launches apart.
        self.logit_fwdbwd = self.prenorm and self.fused_logit_dw and not os.environ.get('SAGE_NO_LOGIT_FWDBWD')
        # The pooling aggregators' last layer the same way (sage_logit_layer_pool_fwd_bwd: neighbour rows
        # from relu(pool(h)), the root row from h, the neighbours' share of the input gradient sent
        # through the relu straight into the gradient of pool(h)): one launch instead of gather /
        # contraction / loss / loss mean / d-input pair / edge pass; the layer's weight gradients and
        # the loss mean ride on the hidden layer's normalise-backward pass, which then works from the
        # pre-activation rows (kept by the forward's normalise pass).  SAGE_NO_FUSED_POOL_LOGIT=1: the
        # separate kernels.
        # every z the layers' contractions add into is handed back zeroed by its consumer: tell the
        # contraction (SAGE_EPI_ZEROED), whose whole-tile shapes then store instead of load-add-store
        # (SAGE_NO_ZEROED_HINT=1: the A/B switch)
        self.zeroed_hint = not os.environ.get('SAGE_NO_ZEROED_HINT')
        self.fused_pool = (self.pool and not unsup and not self.multilabel and len(model.layers) >= 2
                           and ops.logit_layer_supported(last_w.size(1), last_w.size(0))
                           and not os.environ.get('SAGE_NO_FUSED_POOL_LOGIT'))
        self.writers = None         # per sample slot, filled in below once the buffers exist
        self.features, self.labels = features, labels
        self.B = int(batch_size)
        self.L = len(model.layers)
        self.graph = sampler.graph
        self.fanout = list(sampler.num_neighbors)
        self.rng_seed = sampler.seed
        fan = self.fanout
        if len(fan) != self.L or any(k < 0 for k in fan):
            raise NotImplementedError("engine needs one non-negative fan-out per layer")
        dev = features.device
        self.dev = dev
        self._dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        self._cur_seen, self._cur_obj = None, None
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.world_size, self.pg = world_size, process_group
        # capacities: hop_cap[h] = max nodes after hop h
        self.hop_cap = [self.B]
        frontier, edge_cap = self.B, 0
        for k in fan:
            frontier = frontier * k
            edge_cap += frontier
            self.hop_cap.append(self.hop_cap[-1] + frontier)
        # the engine owns its sampler state (relabel table, stamps, buffers): a replayed graph
        # advances the stamp on the device, so the table cannot be shared with eager callers
        self.table = ops.new_relabel_table(self.graph.num_nodes, dev)
        # sample-buffer slots: batch i computes from one while the sampler fills the next; with
        # overlap_gather a third one, because then the sampler runs TWO batches ahead
        if overlap_gather == 'auto':
            overlap_gather = self.gather_pays_overlap(model.layers[0].lin_l.weight.size(1), self.B, fan, unsup)
        self.overlap_gather = bool(overlap_gather)
        # Cross-stream events are the expensive part of pipelining on this platform (a wait and
        # a record on the main stream cost ~20 us per step, the sampler kernels beside the
        # compute ~4): the main stream therefore synchronises with the sampler's stream once per
        # GROUP of `group` batches — the sampler runs `group` batches ahead in a ring of
        # 2*group slots.  overlap_gather needs a per-step hand-shake anyway: 3 slots, group 1.
        self.group = 1 if self.overlap_gather else max(1, int(group))
        self.nslots = 3 if self.overlap_gather else 2 * self.group
        self.bufs = [ops.SampleBuffers(self.graph.num_nodes, self.hop_cap[-1], edge_cap, dev)
                     for _ in range(self.nslots)]
        self.ctrl = torch.zeros(ops.CTRL_WORDS, dtype=torch.int64, device=dev)
        self._stamp = 0
        if (self.logit_fwdbwd or self.fused_pool) and not os.environ.get('SAGE_NO_SOLO_ROWS'):
            # writers per row of the hidden layer's gradient (rows = the targets of layer L-2), per slot
            self.writers = [torch.zeros(self.hop_cap[1], dtype=torch.int32, device=dev) for _ in range(self.nslots)]
        # device-side row counts per slot: n_dev[slot][h] = #nodes after hop h
        self.n_dev = [[b.state[ops.ST_HOP_NODES + h: ops.ST_HOP_NODES + h + 1] for h in range(self.L + 1)]
                      for b in self.bufs]

        self.flat, self.layout = flatten_parameters(model)
        self.grad = torch.zeros_like(self.flat)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        # step counter: Adam call k reads word k&1 and writes word (k+1)&1, so the increment
        # needs no kernel of its own
        self._steps = torch.zeros(2, dtype=torch.int64, device=dev)
        self._adam_calls = 0
        # keep_grads: leave the flat gradient readable after step() (tests); otherwise Adam
        # clears it for the next step and no fill kernel runs
        self.keep_grads = bool(keep_grads)
        self.gviews = dict(zip(model.parameters(), views_like(self.grad, self.layout)))
        # lstm: the forget-gate rows of W_ih and all of W_hh multiply the zero initial state
        # (reference layers/sage.py:187 starts every target's cell from zeros): their gradient is
        # exactly zero at every step, so their moments stay zero and Adam leaves them where they
        # are — without weight decay the optimizer step walks only the other ranges of the flat
        # buffers (27 % less of its 1.4 GB at the Reddit shape).  SAGE_LSTM_ADAM_ALL=1: one launch
        # over everything.
        self.adam_ranges = None
        if self.lstm and weight_decay == 0 and not os.environ.get('SAGE_LSTM_ADAM_ALL'):
            plist = list(model.parameters())
            dead = []
            for layer in model.layers:
                for q, rows in ((layer.lstm.weight_ih, (1, 2)), (layer.lstm.weight_hh, (0, 4))):
                    off, n, shape, ld = self.layout[next(k for k, r in enumerate(plist) if r is q)]
                    fdim = shape[0] // 4
                    dead.append((off + rows[0] * fdim * ld, off + rows[1] * fdim * ld))
            live, pos = [], 0
            for lo, hi in sorted(dead):
                if lo > pos:
                    live.append((pos, lo))
                pos = max(pos, hi)
            if pos < self.flat.numel():
                live.append((pos, self.flat.numel()))
            if all(lo % 4 == 0 and hi % 4 == 0 for lo, hi in live):
                self.adam_ranges = live

        def padded(rows, cols):
            ld = (cols + 15) // 16 * 16
            return torch.zeros((rows, ld), dtype=torch.float32, device=dev)[:, :cols]

        # unsup: the reference trainers' step (trainers/node_level_graphsage_trainers.py:41) adds
        # the unsupervised term — two more forwards over the SAME sampled edges with the features of
        # positive / negative nodes in place of the batch's (trainers/base_trainers.py:170-181) —
        # so every activation exists three times (variant 0 = batch, 1 = positive, 2 = negative)
        # unsup='only': the unsupervised trainer's step (node_level_graphsage_trainers.py:104-117):
        # the same three forwards, loss = the unsupervised term alone (BASELINE.json configs[4])
        self.unsup = bool(unsup)
        self.unsup_only = unsup == 'only'
        # Layer-0 weight gradients without atomics (one GPU, supervised step): the split-K slices of
        # the launch STORE their partial tiles into a workspace of their own and Adam sums them while
        # it reads them (sage_linear_dweight2_parts / sage_adam_step_parts) — fp32 atomics run at a
        # sixth of the plain-store rate and cost that launch 10 of its 64 us.  The
        # flat gradient then holds NOTHING for those two matrices: not under a collective (the
        # all-reduce wants the sums) and not with keep_grads.  SAGE_NO_DW_PARTS=1: atomics.
        self.dw_parts = None
        if (not self.collective and not self.keep_grads and not self.unsup and not self.pool and not self.lstm
                and self.root and not os.environ.get('SAGE_NO_DW_PARTS')):
            params = list(model.parameters())
            l0 = model.layers[0]
            ia, ib = (next(k for k, q in enumerate(params) if q is w) for w in (l0.lin_l.weight, l0.lin_r.weight))
            (oa, na, sha, lda), (ob, nb, shb, ldb) = self.layout[ia], self.layout[ib]
            lo, hi = min(oa, ob), max(oa + na, ob + nb)
            if hi - lo == na + nb and lda is not None and ldb is not None:      # one region of the flat buffer
                max_parts = 32
                ws = torch.zeros(max_parts * (hi - lo), dtype=torch.float32, device=dev)
                self.dw_parts = dict(ws=ws, stride=hi - lo, off=lo, len=hi - lo, max=max_parts, n=0,
                                     holder=ctypes.c_int(0),      # (written by every replay of the recorded call)
                                     view_l=ws[oa - lo:oa - lo + na].view(sha[0], lda)[:, :sha[1]],
                                     view_r=ws[ob - lo:ob - lo + nb].view(shb[0], ldb)[:, :shb[1]])
        nvar = 3 if self.unsup else 1

        def make_act():
            act = []
            for i, layer in enumerate(model.layers):
                t_cap = self.hop_cap[self.L - 1 - i]
                fin = layer.lin_l.weight.size(1)
                fout = layer.lin_l.weight.size(0)
                a = dict(t_cap=t_cap, fin=fin, fout=fout, hop=self.L - 1 - i,
                         agg=padded(t_cap, fin), z=padded(t_cap, fout), y=padded(t_cap, fout),
                         inv=torch.zeros(t_cap, dtype=torch.float32, device=dev),
                         dz=padded(t_cap, fout))
                if i < self.L - 1:
                    a['h'] = padded(t_cap, fout)
                    a['dh'] = padded(t_cap, fout)
                else:
                    a['dy'] = padded(t_cap, fout)     # unsup: gradient w.r.t. the normalised logits
                if self.lstm:
                    n_in = int(layer.lstm_n_inputs)
                    a['n_in'] = n_in
                    a['xflat'] = padded(t_cap, n_in * fin)      # the cell's flattened input, one row per target
                    a['gates'] = padded(t_cap, 4 * fin)         # x_flat . W_ih^T (biases are added by the cell kernels)
                    a['dgates'] = padded(t_cap, 4 * fin)
                    a['nonempty'] = torch.zeros(t_cap, dtype=torch.float32, device=dev)
                    if i > 0:
                        a['dxflat'] = padded(t_cap, n_in * fin)
                    # Short segments: the rows behind the seeds (layer 0: the hop-1 nodes and beyond) got
                    # their edges from a hop with fan-out <= kpad, so their slots kpad .. n_in-1 all
                    # repeat the first neighbour (reference layers/sage.py:164) and contract as ONE slot
                    # against the sum of those slots' weight blocks: kpad + 1 slot-products instead of
                    # n_in for 96 % of the Reddit batch's targets (k2 = 10, n_in = 25).  Layer 0 only
                    # (no input gradient there); SAGE_LSTM_NO_PADSUM=1 keeps the full contraction.
                    kpad = max(fan[1:]) if len(fan) > 1 else n_in      # (fan-outs are non-negative here)
                    if (i == 0 and t_cap > self.B and kpad < n_in and not os.environ.get('SAGE_LSTM_NO_PADSUM')):
                        a['kpad'] = kpad
                        a['wsum'] = padded(4 * fin, fin)          # sum of the padded slots' weight blocks
                        a['gpad'] = padded(4 * fin, fin)          # its gradient (zero between steps)
                        a['rest_dev'] = torch.zeros(1, dtype=torch.int64, device=dev)   # live targets behind the seeds
                if i > 0 or self.pool or self.lstm:
                    a['dagg'] = padded(t_cap, fin)
                    if ((self.lstm or self.pool) and i == 0 and t_cap >= 4096 and fin >= 256
                            and not os.environ.get('SAGE_NO_DAGG_X3')):
                        # layer 0's d agg = dz . W_l as a FORWARD-type contraction on a transposed copy
                        # of W_l (both operands contiguous in the contracted index: the bf16x3 kernel
                        # takes it; the d-input kernel is exact-fp32 only, 55-69 us at the Reddit shape):
                        # lstm 9.12 against 9.18 ms per step.  For the pooling aggregators, where this
                        # contraction runs beside the forked weight gradients, the same-box A/B first went
                        # the other way (max_pool 1.209 against 1.200 ms with the pushed edge pass behind
                        # it, scratch/dagg_ab.sh); with the pulled edge pass and the fused last layer it
                        # pays: max_pool 1.143 against 1.156, mean_pool 1.183 against 1.197
                        # (scratch/pool_dagg_ab2.sh, profiles/r3_pool_dagg_ab.txt; SAGE_NO_DAGG_X3=1)
                        a['wlT'] = padded(fin, fout)
                    if self.reduce == 'max':   # the backward routes to the edge that attained the max
                        a['argmax'] = torch.zeros((t_cap, fin), dtype=torch.int64, device=dev)
                if self.pool:
                    # sources of layer i = the nodes after hop L-i (all sampled nodes for layer 0)
                    a['src_hop'] = self.L - i
                    s_cap = self.hop_cap[self.L - i]
                    a['s_cap'] = s_cap
                    a['P'] = padded(s_cap, fin)       # relu(pool(x)) per source row
                    # gradient w.r.t. pool(x), zeroed by the step's pool Linear and filled by the
                    # masked scatter (sage_pool_max_bwd / sage_pool_mean_bwd: no dense d P)
                    a['dpre'] = padded(s_cap, fin)
                    a['indeg'] = torch.zeros(s_cap, dtype=torch.int32, device=dev)   # the batch's edges per source row
                act.append(a)
            return act
        self.acts = [make_act() for _ in range(nvar)]
        self.act = self.acts[0]   # per layer: dict of static buffers (variant 0)
        # Pooling aggregators, layer 0: the edge pass of the backward PULLED by the source rows
        # (sage_pool_max_bwd_pull / sage_pool_mean_bwd_pull: every row of d pre stored once, no
        # atomics, no in-degree launches in the step, no zero-fill by the pool Linear) from the
        # batch's edges listed per source row, which the sampler's stream builds behind the sampler
        # (sage_pool_reverse_edges, per sample slot); max keeps its arg-max as one byte per (target,
        # column).  (The layers above: mean_pool's take the same pass, rev_up below; the last layer
        # is the fused launch, fused_pool.)
        # SAGE_NO_POOL_PULL=1: the pushed pass with its in-degree workspace for layer 0 too;
        # SAGE_POOL_PULL_ALL=1: pulled whatever the batch's size (below 4096 targets the lists cost
        # the sampler's stream more than the step gains; tests).
        self.rev = None
        self._p_all = None          # evaluate(): relu(pool(x)) over all nodes while a pass runs
        self._eval_pass = False     # evaluate(): the sampler's tape without the helpers of the backward
        self._p_all_buf = None
        a0 = self.act[0]
        if (self.pool and (a0['t_cap'] >= 4096 or os.environ.get('SAGE_POOL_PULL_ALL')) and a0['t_cap'] < 2 ** 23
                and not os.environ.get('SAGE_NO_POOL_PULL')
                and (self.reduce != 'max' or max(fan) <= 255)):
            self.rev = [ops.ReverseEdges(a0['s_cap'], edge_cap, dev) for _ in range(self.nslots)]
            if self.reduce == 'max':
                a0['arg8'] = torch.zeros((a0['t_cap'], (a0['fin'] + 15) // 16 * 16), dtype=torch.uint8, device=dev)
        # mean_pool, the layers above layer 0 (the B seeds' segments): the same pulled pass instead of
        # the all-atomic one (17 us -> 5 at the Reddit shape), lists per layer and slot.  Not for max
        # there: its pushed pass sends one value per (seed, column), 5 us.
        self.rev_up = None
        if self.rev is not None and self.reduce == 'mean' and self.L >= 2 and not os.environ.get('SAGE_NO_POOL_PULL_UP'):
            self.rev_up = {i: [ops.ReverseEdges(self.act[i]['s_cap'], self.act[i]['t_cap'] * max(fan), dev)
                               for _ in range(self.nslots)]
                           for i in range(1, self.L) if not (self.fused_pool and i == self.L - 1)}
        if self.unsup:
            self.pos_id = torch.zeros(self.hop_cap[-1], dtype=torch.int64, device=dev)
            self.neg_id = torch.zeros(self.hop_cap[-1], dtype=torch.int64, device=dev)
            self.loss_sup = torch.zeros(1, dtype=torch.float32, device=dev)
            self.loss_unsup = torch.zeros(1, dtype=torch.float32, device=dev)
            self.slot_counter = [0] * self.nslots
        # layer-0 aggregate per sample slot: with overlap_gather the HBM-bound gather of batch i+1
        # runs on the side stream (it needs no weights) beside the MFMA-bound forward GEMM of
        # batch i, into a buffer of its own
        self.agg0 = [self.act[0]['agg']] * self.nslots
        if self.overlap_gather:
            self.agg0 = [self.act[0]['agg']] + [padded(self.act[0]['t_cap'], self.act[0]['fin'])
                                                 for _ in range(self.nslots - 1)]
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.row_loss = torch.zeros(self.B, dtype=torch.float32, device=dev)
        # running sums of the sampler's state block over SAMPLED batches (batch sizes; a non-zero
        # ST_ERROR sum means some batch reported an error); accumulated on the sampler's stream
        self.state_sum = torch.zeros(ops.STATE_WORDS, dtype=torch.int64, device=dev)
        self.counter = 0
        self.graphs = None
        self._tapes = {}        # recorded C-ABI call sequences, keyed by (what, slot, ..., stream)
        self.use_tapes = True   # False: every call goes through the validating wrappers of ops.py (measurement hooks)
        self._pending = PendingReduce()   # data parallel: the gradient all-reduce not yet applied
        self.trace = None       # tests: a list that receives the name of every piece of a step as it is ENQUEUED
        self._side_first = bool(os.environ.get('SAGE_SIDE_FIRST'))   # A/B: the round-2 enqueue order
        # data parallel: the host may run at most `run_ahead` steps (= un-waited collectives)
        # ahead of the device: it blocks on the event recorded that many steps ago
        self.run_ahead = 8
        self._dp_events = []
        self._dp_ring = [torch.cuda.Event() for _ in range(3)]
        # pipelining state
        main = torch.cuda.current_stream(dev)
        self.side = new_side_stream(dev, [main])
        # overlap_gather: the sampler gets a stream of its own.  Its short latency-bound kernels
        # only find free CU resources between the big kernels (a resident bf16x3 GEMM holds every
        # VGPR), and on the gather's stream that wait would delay the gather the main stream needs
        # next; the sampler's result is needed two batches later.
        self.samp = new_side_stream(dev, [main, self.side]) if self.overlap_gather else self.side
        self.pool_fork = new_side_stream(dev, [main]) if (self.pool and not os.environ.get('SAGE_NO_POOL_FORK')) else None
        # data parallel: the gradient all-reduce runs on a stream of the engine's own, on a hardware
        # queue that is neither the main stream's nor the sampler's (SAGE_DP_BACKEND_STREAM=1: leave
        # the stream to the backend — the asynchronous work-handle path)
        self.comm = (new_side_stream(dev, [main, self.side]) if (self.collective and not os.environ.get('SAGE_DP_BACKEND_STREAM'))
                     else None)
        # the three-forward steps: the positive and the negative variant run on streams of their own
        # beside the batch's (kernels of 512 to 13 k rows leave most of the chip idle one at a time);
        # fork behind make_pos_neg, join in front of the loss, fork again for the three backward
        # passes (their weight gradients meet in fp32 atomics either way), join in front of Adam
        # (SAGE_NO_UNSUP_FORK=1: one stream, the A/B switch)
        self.var_streams = None
        if self.unsup and not os.environ.get('SAGE_NO_UNSUP_FORK'):
            v1 = new_side_stream(dev, [main, self.side])
            self.var_streams = [v1, new_side_stream(dev, [main, self.side, v1])]
            # fork and join through events of the engine's own, recorded and waited for by library
            # calls on the step's tape (one record + two waits to fork, two + two to join: ~10 us of
            # host time per step for both phases; four stream-to-stream waits of the host framework
            # per phase were 120 us and left the products-scale step host-bound)
            self.var_ev = [ops.new_event(main), ops.new_event(v1), ops.new_event(self.var_streams[1])]
        self._forked = False
        self.sampled = [torch.cuda.Event() for _ in range(self.nslots)]     # slot sampled (sampler stream)
        self.ready = [torch.cuda.Event() for _ in range(self.nslots)]   # slot sampled (side stream)
        self.cons_ev = [torch.cuda.Event(), torch.cuda.Event()]   # main stream finished a group
        for ev in self.cons_ev:
            ev.record()             # (creates the handle: the Adam launch can carry it as its completion event)
        self._adam_signal = not os.environ.get('SAGE_NO_ADAM_SIGNAL')
        self.started = torch.cuda.Event()    # main stream reached the start of a step
        self._gate = self.started
        self._own_gate = bool(os.environ.get('SAGE_STARTED_EVENT'))
        self.n_pref = 0         # batches prefetched so far; batch n lives in slot n % nslots
        self.n_comp = 0         # batches computed so far
        self.main_ready_seq = -1   # newest batch whose sampling the main stream has waited for
        self.cons_rec_seq = -1     # newest batch covered by a recorded cons_ev
        self.side_cons_seq = -1    # newest batch whose compute the side stream has waited for
        self.cons_latest = None
        self.consumed = [torch.cuda.Event(), torch.cuda.Event()]   # graph replay: slot's compute finished
        self._main_touched = True   # the main stream used sample buffers outside the pipeline (step/evaluate)
        self.gathered = [False] * self.nslots   # overlap_gather: slot's layer-0 aggregate is done
        self.use = 0            # graph replay: slot the next compute reads

    @property
    def primed(self):
        return self.n_pref > self.n_comp

    @property
    def depth(self):
        """How many batches ahead of the one being computed the sampler runs: callers prime the
        pipeline with `depth` prefetch() calls and hand pipelined_step() batch i + depth."""
        return 2 if self.overlap_gather else self.group

    @property
    def step_dev(self):
        """Number of optimizer steps taken (device tensor)."""
        return self._steps.max()

    @property
    def buf(self):
        """Sample buffers of the batch computed last (slot 0 in the non-pipelined mode)."""
        return self.bufs[self.last_slot] if hasattr(self, 'last_slot') else self.bufs[0]

    # ------------------------------------------------------------------ pieces
    def sample(self, seeds, counter, slot=0, ctrl=None, stream=None):
        """Enqueue the sampler for the batch `seeds[offset : offset+B]` (offset from ctrl or 0)
        on `stream` (a torch stream; default: the current one)."""
        if ctrl is not None:       # graph capture: the device control block carries the running stamp
            ops.sample_fanout(self.graph.colptr, self.graph.row, seeds, self.fanout, self.rng_seed, counter,
                              self.table, 1, self.bufs[slot], ctrl=ctrl, num_seeds=self.B)
            ops.state_accumulate(self.bufs[slot].state, self.state_sum)
            self._count_writers(slot)
            return
        self._next_stamp(stream)
        if self.unsup:
            self.slot_counter[slot] = counter
        handle = stream.cuda_stream if stream is not None else self._cur_raw()
        key = ('sample', slot, handle)
        entry = self._tapes.get(key)
        if entry is None:
            # first call on this (slot, stream): through the validating wrapper, recorded
            def first():
                ops.sample_fanout(self.graph.colptr, self.graph.row, seeds, self.fanout, self.rng_seed,
                                  counter, self.table, self._stamp, self.bufs[slot], num_seeds=self.B)
                ops.state_accumulate(self.bufs[slot].state, self.state_sum)
                self._count_writers(slot)
            if stream is not None and stream != torch.cuda.current_stream():
                with torch.cuda.stream(stream), _lib.recording() as tape:
                    first()
            else:
                with _lib.recording() as tape:
                    first()
            (fn, args, name), acc = tape[:2]
            self._tapes[key] = (fn, list(args), name, acc, tape[2:])
            return
        # later calls: same pointers except the seeds, the Philox counter and the stamp
        if not (seeds.is_cuda and seeds.dtype == torch.int64 and seeds.dim() == 1 and seeds.is_contiguous()
                and seeds.numel() >= self.B):
            raise RuntimeError("sample: seeds must be a contiguous CUDA int64 vector of at least B elements")
        fn, args, name, acc, rest = entry
        args[3] = seeds.data_ptr()
        args[8] = int(counter) & 0xFFFFFFFFFFFFFFFF
        args[10] = self._stamp
        _lib.check(fn(*args), name)
        _lib.check(acc[0](*acc[1]), acc[2])      # state_sum += this batch's sizes, same stream
        if rest and not self._eval_pass:
            _lib.replay(rest)                    # (the backward's helpers: writer counts, per-source edge lists)

    def _count_writers(self, slot):
        """Behind the sampler, on its stream: how many seed waves will add into each row of the
        hidden layer's gradient (sage_logit_layer_fwd_bwd stores the rows with one writer)."""
        if self.writers is not None:
            buf = self.bufs[slot]
            ops.count_row_writers(buf.seg_ptr, buf.src, self.B, self.writers[slot], max_edges=self.B * max(self.fanout[0], 1))
        if self.rev is not None:      # (pooling) layer 0's edges per source row, for its backward's edge pass
            a, buf = self.act[0], self.bufs[slot]
            self.rev[slot].build(buf.seg_ptr, buf.src, a['t_cap'], num_targets_dev=self.n_dev[slot][a['hop']])
            if self.rev_up is not None:
                for i, revs in self.rev_up.items():
                    a = self.act[i]
                    revs[slot].build(buf.seg_ptr, buf.src, a['t_cap'], num_targets_dev=self.n_dev[slot][a['hop']])

    def _next_stamp(self, stream=None):
        """Stamps of the relabel table must grow (sage_hip.h); when the 32-bit range is used up,
        start over with a clean table (as graphsage_amd/samplers/uniform_loader.py does)."""
        self._stamp += 1
        if self._stamp >= 2 ** 32 - 1:
            if stream is not None and stream != torch.cuda.current_stream():
                with torch.cuda.stream(stream):
                    self.table.fill_(-1)
            else:
                self.table.fill_(-1)
            self._stamp = 1

    def _require_eager(self, what):
        """After capture() the graphs own the sampler state: a replay advances the stamp and the
        Philox counter on the DEVICE (control block) and always leaves a sampler graph in flight
        on the side stream, so an eager call would sample with stale stamps into buffers that a
        graph is writing.  Everything eager refuses; release_graphs() hands the state back."""
        if self.graphs is not None:
            raise RuntimeError(f"{what}: the engine was captured into HIP graphs — use replay(), or "
                               "release_graphs() first")

    def release_graphs(self):
        """Leave graph mode: wait for the graphs in flight and take the stamp they advanced on the
        device back to the host, so that the eager entry points continue above it."""
        if self.graphs is None:
            return
        torch.cuda.synchronize()
        self._stamp = int(self.ctrl[ops.CTRL_STAMP]) + 1      # captured with host stamp 1 + ctrl
        self.graphs = None
        self._tapes.clear()        # (recorded again on the next eager call: one stream, no fork)
        self._main_touched = True

    def _cur_raw(self):
        """Raw handle of torch's current stream on the engine's device (torch.cuda.current_stream()
        spends ~9 us of host time on device-index bookkeeping: four calls per step)."""
        return torch._C._cuda_getCurrentRawStream(self._dev_index)

    def _cur(self):
        """torch.cuda.current_stream() of the engine's device, cached by its raw handle."""
        raw = torch._C._cuda_getCurrentRawStream(self._dev_index)
        if raw != self._cur_seen:
            with torch.cuda.device(self._dev_index):
                self._cur_obj = torch.cuda.current_stream()
            self._cur_seen = raw
        return self._cur_obj

    def _taped(self, what, slot, fn, stream=None):
        """Run fn() through the validating wrappers once per (what, slot, stream) and replay the
        recorded C-ABI calls afterwards: every pointer the engine passes is static.  `stream`
        (default: the current one) is where the calls go; a replay needs no stream switch."""
        cur = self._cur()
        if not self.use_tapes:
            if stream is not None and stream != cur:
                with torch.cuda.stream(stream):
                    fn()
            else:
                fn()
            return
        key = (what, slot, (stream or cur).cuda_stream)
        tape = self._tapes.get(key)
        if tape is None:
            if stream is not None and stream != cur:
                with torch.cuda.stream(stream), _lib.recording() as tape:
                    fn()
            else:
                with _lib.recording() as tape:
                    fn()
            self._tapes[key] = tape
        else:
            _lib.replay(tape)

    def _note(self, what):
        if self.trace is not None:
            self.trace.append(what)

    def gather_layer0(self, slot=0, stream=None, events=None):
        """Layer-0 gather of the batch in `slot`.  events = (start, stop) raw HIP event handles:
        the launch carries them (sage_gather_reduce_fwd_timed), for a caller that measures this
        kernel (bench.py's roofline probe)."""
        self._note('gather0')
        if events is None:
            self._taped('gather0' if self._p_all is None else 'gather0_all', slot, lambda: self._gather_layer0(slot), stream)
            return
        if self.pool:
            raise NotImplementedError("gather_layer0(events=...): with a pooling aggregator the step's first "
                                      "kernel is the pool contraction, not the gather")
        cur = torch.cuda.current_stream()
        key = ('gather0t', slot, (stream or cur).cuda_stream)
        entry = self._tapes.get(key)
        if entry is None:
            if stream is not None and stream != cur:
                with torch.cuda.stream(stream), _lib.recording() as tape:
                    self._gather_layer0(slot, events)
            else:
                with _lib.recording() as tape:
                    self._gather_layer0(slot, events)
            (fn, args, name), = tape
            self._tapes[key] = (fn, list(args), name)
            return
        fn, args, name = entry
        args[12], args[13] = int(events[0]), int(events[1])     # same pointers, this step's events
        _lib.check(fn(*args), name)

    def _gather_layer0(self, slot, events=None):
        a, buf = self.act[0], self.bufs[slot]
        if self.lstm:
            if events is not None:
                raise NotImplementedError("gather_layer0(events=...): the lstm aggregate is a contraction, not a gather")
            self._lstm_forward(slot, 0, self.features, buf.node, self.agg0[slot])
            return
        if self.pool and self._p_all is not None:
            # evaluate(): relu(pool(x)) of EVERY node was computed once for the pass (the weights
            # stand still), the batch only gathers its rows of it
            ops.gather_reduce_fwd(self._p_all, buf.seg_ptr, buf.src, a['t_cap'], self.reduce, node_map=buf.node,
                                  num_targets_dev=self.n_dev[slot][a['hop']], out=self.agg0[slot], want_argmax=False)
            return
        if self.pool:
            self._pool_forward(slot, 0, self.features, buf.node)
            if 'arg8' in a:
                ops.pool_max_fwd(a['P'], buf.seg_ptr, buf.src, a['t_cap'], max(self.fanout), self.agg0[slot], a['arg8'],
                                 num_targets_dev=self.n_dev[slot][a['hop']])
                return
            ops.gather_reduce_fwd(a['P'], buf.seg_ptr, buf.src, a['t_cap'], self.reduce,
                                  num_targets_dev=self.n_dev[slot][a['hop']], out=self.agg0[slot],
                                  want_argmax=self.reduce == 'max', argmax=a.get('argmax'), events=events)
            return
        ops.gather_reduce_fwd(self.features, buf.seg_ptr, buf.src, a['t_cap'], self.reduce,
                              node_map=buf.node, num_targets_dev=self.n_dev[slot][a['hop']],
                              out=self.agg0[slot], want_argmax=False, events=events)

    def _pool_forward(self, slot, i, x, node_map):
        """P = relu(x[rows] @ pool.weight^T + pool.bias) over the layer's source rows."""
        a, layer = self.act[i], self.model.layers[i]
        # the same stores hand dpre back zeroed for this step's masked scatter (the pulled pass of
        # layer 0 stores every row itself)
        ops.pool_linear_fwd(x, layer.pool.weight, layer.pool.bias, m=a['s_cap'],
                            m_dev=self.n_dev[slot][a['src_hop']], rows=node_map, out=a['P'],
                            dpre=None if ((i == 0 and self.rev is not None) or
                                          (i > 0 and self.rev_up is not None and i in self.rev_up))
                            else a['dpre'])

    @staticmethod
    def _lstm_blocks(fdim):
        """Row blocks of W_ih (torch order i, f, g, o) that reach the output from a zero state: the
        forget gate multiplies c0 = 0 and is neither computed nor given a gradient."""
        return ((0, fdim), (2 * fdim, 4 * fdim))

    def _lstm_forward(self, slot, i, x, node_map, agg):
        """agg[t] = LSTMCell(x_flat[t]) from a zero state for the layer's targets."""
        a, layer, buf = self.act[i], self.model.layers[i], self.bufs[slot]
        t_cap, t_dev, fdim = a['t_cap'], self.n_dev[slot][a['hop']], a['fin']
        w = layer.lstm.weight_ih
        kp = a.get('kpad')
        if kp is None:
            ops.lstm_gather_flat(x, buf.seg_ptr, buf.src, t_cap, a['n_in'], a['xflat'], a['nonempty'],
                                 node_map=node_map, num_targets_dev=t_dev)
            for r0, r1 in self._lstm_blocks(fdim):
                ops.linear_concat_fwd(a['xflat'], w[r0:r1], m=t_cap, m_dev=t_dev, out=a['gates'][:, r0:r1])
        else:
            B, rest_dev, xf = self.B, a['rest_dev'], a['xflat']
            _lib.tape_py(lambda: torch.sub(t_dev, B, out=rest_dev))
            # (slots kp .. n_in-1 of the rows behind the seeds are never read: not written either)
            ops.lstm_gather_flat(x, buf.seg_ptr, buf.src, B, a['n_in'], xf, a['nonempty'], node_map=node_map)
            ops.lstm_gather_flat(x, buf.seg_ptr[B:], buf.src, t_cap - B, kp, xf[B:, :kp * fdim], a['nonempty'][B:],
                                 node_map=node_map, num_targets_dev=rest_dev)
            ops.lstm_pad_weight_sum(w, fdim, kp, a['n_in'], a['wsum'])
            for r0, r1 in self._lstm_blocks(fdim):
                # the seeds: up to n_in real neighbours, the full contraction
                ops.linear_concat_fwd(xf[:B], w[r0:r1], m=B, out=a['gates'][:B, r0:r1])
                # the rows behind them: kp slots + slot 0 once more against the summed weights
                ops.linear_concat_fwd(xf[B:, :kp * fdim], w[r0:r1, :kp * fdim], xf[B:, :fdim], a['wsum'][r0:r1],
                                      m=t_cap - B, m_dev=rest_dev, out=a['gates'][B:, r0:r1])
        ops.lstm_cell0_fwd(a['gates'], a['nonempty'], fdim, m=t_cap, m_dev=t_dev, bias_a=layer.lstm.bias_ih,
                           bias_b=layer.lstm.bias_hh, out=agg)

    def _lstm_backward(self, slot, i):
        """d agg -> (cell) d gates -> W_ih / bias gradients and, above layer 0, the input gradient
        scattered into the previous layer's dh."""
        a, layer, buf = self.act[i], self.model.layers[i], self.bufs[slot]
        t_cap, t_dev, fdim = a['t_cap'], self.n_dev[slot][a['hop']], a['fin']
        if i > 0:      # root term into the previous layer's dh (overwrites its first t rows), d agg beside it
            ops.linear_dinput2(a['dz'], layer.lin_r.weight, self.act[i - 1]['dh'], layer.lin_l.weight, a['dagg'],
                               m=t_cap, m_dev=t_dev)
        else:
            self._dagg_layer0(slot)
        ops.lstm_cell0_bwd(a['gates'], a['nonempty'], a['dagg'], fdim, m=t_cap, m_dev=t_dev,
                           bias_a=layer.lstm.bias_ih, bias_b=layer.lstm.bias_hh, out=a['dgates'])
        gw, gb_ih, gb_hh = (self.gviews[p] for p in (layer.lstm.weight_ih, layer.lstm.bias_ih, layer.lstm.bias_hh))
        kp = a.get('kpad')
        if kp is None:
            for r0, r1 in self._lstm_blocks(fdim):
                ops.linear_dweight(a['dgates'][:, r0:r1], a['xflat'], m=t_cap, m_dev=t_dev, dw=gw[r0:r1],
                                   dbias=gb_ih[r0:r1])
        else:      # (see _lstm_forward: the same split, differentiated)
            B, rest_dev, xf, dg = self.B, a['rest_dev'], a['xflat'], a['dgates']
            for r0, r1 in self._lstm_blocks(fdim):
                ops.linear_dweight(dg[:B, r0:r1], xf[:B], m=B, dw=gw[r0:r1], dbias=gb_ih[r0:r1])
                ops.linear_dweight(dg[B:, r0:r1], xf[B:, :kp * fdim], m=t_cap - B, m_dev=rest_dev,
                                   dw=gw[r0:r1, :kp * fdim], dbias=gb_ih[r0:r1])
                ops.linear_dweight(dg[B:, r0:r1], xf[B:, :fdim], m=t_cap - B, m_dev=rest_dev, dw=a['gpad'][r0:r1])
            ops.lstm_pad_grad_bcast(a['gpad'], fdim, kp, a['n_in'], gw)
        _lib.tape_py(lambda: gb_hh.copy_(gb_ih))       # b_ih and b_hh enter the gates as their sum
        if i > 0:
            w = layer.lstm.weight_ih
            (a0, a1), (b0, b1) = self._lstm_blocks(fdim)
            ops.linear_dinput(a['dgates'][:, a0:a1], w[a0:a1], m=t_cap, m_dev=t_dev, out=a['dxflat'])
            ops.linear_dinput(a['dgates'][:, b0:b1], w[b0:b1], m=t_cap, m_dev=t_dev, out=a['dxflat'], accumulate=True)
            ops.lstm_scatter_flat(a['dxflat'], buf.seg_ptr, buf.src, t_cap, a['n_in'], self.act[i - 1]['dh'],
                                  num_targets_dev=t_dev)

    def forward_backward(self, slot=0, skip_gather0=False):
        self._note('forward_backward')
        self.last_slot = slot
        if self.unsup:
            self.make_pos_neg(slot, self.slot_counter[slot])

        def both():
            self._forward(slot, skip_gather0, fused=self.fused_logit or self.fused_pool)
            self._backward(slot, fused=self.fused_logit or self.fused_pool)
        self._taped('fb%d' % skip_gather0, slot, both)

    def forward(self, slot=0, skip_gather0=False):
        """Layers + normalised logits + per-row cross entropy (and dz of the last layer, which
        the fused kernel produces in the same pass)."""
        self.last_slot = slot
        if self.unsup:
            self.make_pos_neg(slot, self.slot_counter[slot])
        self._taped('f%d' % skip_gather0, slot, lambda: self._forward(slot, skip_gather0))

    def backward(self, slot=0):
        self._taped('b', slot, lambda: self._backward(slot))

    def _node_map(self, slot, v):
        return (self.bufs[slot].node, self.pos_id, self.neg_id)[v] if self.unsup else self.bufs[slot].node

    def _agg0(self, slot, v):
        return self.agg0[slot] if v == 0 else self.acts[v][0]['agg']

    def _forward_layers(self, slot, v, skip_gather0, upto=None, raw_last=False, keep_z_last=False):
        """Layers of variant v up to the un-normalised logits z (or only the first `upto`;
        raw_last: the last of those is left as its pre-activation rows z — the fused class-logit
        kernel normalises what it reads)."""
        L, buf = self.L, self.bufs[slot]
        act = self.acts[v]
        x, node_map = self.features, self._node_map(slot, v)
        lib = ops
        for i, layer in enumerate(self.model.layers):
            if upto is not None and i >= upto:
                break
            a = act[i]
            t_cap, t_dev = a['t_cap'], self.n_dev[slot][a['hop']]
            agg = self._agg0(slot, v) if i == 0 else a['agg']
            if i == 0:
                if v > 0:
                    lib.gather_reduce_fwd(x, buf.seg_ptr, buf.src, t_cap, self.reduce, node_map=node_map,
                                          num_targets_dev=t_dev, out=agg, want_argmax=False)
                elif not skip_gather0:
                    self._gather_layer0(slot)
            elif self.lstm:
                self._lstm_forward(slot, i, x, None, agg)
            else:
                if self.pool:
                    self._pool_forward(slot, i, x, None)
                lib.gather_reduce_fwd(a['P'] if self.pool else x, buf.seg_ptr, buf.src, t_cap, self.reduce,
                                      node_map=None, num_targets_dev=t_dev, out=agg,
                                      want_argmax=self.reduce == 'max', argmax=a.get('argmax'))
            # The GEMMs ADD into z (stream-K / split-K partial sums), and every z starts as zeros:
            # the consumer hands it back zeroed (clear_z of the normalise kernel, of the fused
            # loss kernel in the supervised step, of the unsup variants' normalise pass), so there is
            # no fill kernel.
            lib.linear_concat_fwd(agg, layer.lin_l.weight, x if self.root else None,
                                  layer.lin_r.weight if self.root else None, m=t_cap,
                                  m_dev=t_dev, rows2=node_map if self.root else None, out=a['z'],
                                  accumulate='zeroed' if self.zeroed_hint else True)
            if i < L - 1 and not (raw_last and i == upto - 1):
                # (keep_z_last: the backward of that layer works from z, sage_l2norm_relu_bwd_z_dw, and clears it)
                lib.l2norm_relu_fwd(a['z'], True, m=t_cap, m_dev=t_dev, y=a['y'], h=a['h'], inv_norm=a['inv'],
                                    clear_z=not (keep_z_last and i == upto - 1))
            x, node_map = a.get('h'), None

    def _forward(self, slot, skip_gather0, fused=False):
        buf, lib, B = self.bufs[slot], ops, self.B
        if fused and self.fused_pool:
            self._forward_layers(slot, 0, skip_gather0, upto=self.L - 1, keep_z_last=True)
            last, layer, prev = self.act[-1], self.model.layers[-1], self.act[-2]
            self._pool_forward(slot, self.L - 1, prev['h'], None)       # P = relu(pool(h)); d pre handed back zeroed
            lib.logit_layer_pool_fwd_bwd(prev['h'], last['P'], buf.seg_ptr, buf.src, B, self.reduce, layer.lin_l.weight,
                                         layer.lin_r.weight, self.labels, buf.node, last['agg'], last['y'], last['inv'],
                                         last['dz'], self.row_loss, prev['dh'], last['dpre'],
                                         writers=self.writers[slot] if self.writers is not None else None)
            return
        if fused:
            # (training step only: the loss mean then comes out of the backward launch)
            self._forward_layers(slot, 0, skip_gather0, upto=self.L - 1, raw_last=self.prenorm)
            last, layer, prev = self.act[-1], self.model.layers[-1], self.act[-2]
            # prenorm: no normalise pass over the hidden layer (54 MB and a launch) — the kernel
            # takes the ~11 rows per seed it reads through relu(normalize(.)) in registers and
            # leaves the seeds' own activations in h for the root term's weight gradient; the
            # backward works from z as well (sage_l2norm_relu_bwd_z) and hands it back zeroed
            # (gcn: no root weight — the kernel takes w_l in its place and does not use it)
            lib.logit_layer_fwd(prev['z'] if self.prenorm else prev['h'], buf.seg_ptr, buf.src, B, self.reduce,
                                layer.lin_l.weight, layer.lin_r.weight if self.root else layer.lin_l.weight,
                                self.labels, buf.node, last['agg'], last['y'],
                                last['inv'], last['dz'], self.row_loss, h_root=prev['h'] if self.prenorm else None,
                                dh=prev['dh'] if self.logit_fwdbwd else None,
                                writers=self.writers[slot] if (self.logit_fwdbwd and self.writers is not None) else None)
            return
        if not self.unsup:
            self._forward_layers(slot, 0, skip_gather0)
            # last layer: normalise + loss on the seed rows + gradient w.r.t. z, one kernel
            self._loss_last(slot, B)
            return
        for v in (1, 2, 0):      # (the forked variants first: they start while the host enqueues the batch's)
            with self._variant(v):
                self._forward_layers(slot, v, skip_gather0)
                last = self.acts[v][-1]
                lib.l2norm_relu_fwd(last['z'], False, m=B, y=last['y'], inv_norm=last['inv'], clear_z=True)
        self._join_variants()
        l0, l1, l2 = (self.acts[v][-1] for v in range(3))
        # loss = CE(out, y) + unsup(out, pos_out, neg_out)   (node_level_graphsage_trainers.py:41)
        if self.unsup_only:
            dy0 = l0['dy'][:B]
            _lib.tape_py(dy0.zero_)               # sage_unsup_loss ADDS the batch rows' gradient
            lib.unsup_loss(l0['y'], l1['y'], l2['y'], l0['dy'], l1['dy'], l2['dy'], self.row_loss, self.loss, m=B)
        else:
            lib.softmax_xent(l0['y'], self.labels, label_rows=buf.node, m=B, loss=self.loss_sup, dlogits=l0['dy'],
                             checked=True, row_loss=self.row_loss)     # (unsup_loss reuses row_loss behind it)
            lib.unsup_loss(l0['y'], l1['y'], l2['y'], l0['dy'], l1['dy'], l2['dy'], self.row_loss, self.loss_unsup,
                           m=B)
            _lib.tape_py(lambda: torch.add(self.loss_sup, self.loss_unsup, out=self.loss))
        # (d y -> d z of the three logit layers: the first kernel of each variant's backward pass)

    def _loss_last(self, slot, m):
        """normalise + loss over the first m seed rows + d loss / d z: cross entropy on class ids
        (fig2a/runners/graphsage_runners.py:43) or BCE-with-logits on multi-label targets."""
        buf, last = self.bufs[slot], self.act[-1]
        if self.multilabel:
            ops.logits_bce_fused(last['z'], self.labels, buf.node, m, y=last['y'], inv_norm=last['inv'], dz=last['dz'],
                                 row_loss=self.row_loss, loss=self.loss, clear_z=True)
        else:
            ops.logits_xent_fused(last['z'], self.labels, buf.node, m, last['y'], last['inv'], last['dz'],
                                  self.row_loss, self.loss, clear_z=True)

    def _backward(self, slot, fused=False):
        if self.keep_grads:
            _lib.tape_py(self.grad.zero_)
        if self.unsup:
            for v in (1, 2, 0):
                with self._variant(v):
                    last = self.acts[v][-1]
                    ops.l2norm_relu_bwd(last['dy'], last['y'], last['inv'], False, m=self.B, dz=last['dz'])
                    self._backward_layers(slot, v, fused)
            self._join_variants()
        else:
            self._backward_layers(slot, 0, fused)
        if self._forked:      # everything behind the backward (Adam, the all-reduce) sees the forked gradients
            main = torch.cuda.current_stream()
            _lib.tape_py(lambda: main.wait_stream(self.pool_fork))
            self._forked = False

    def _backward_layers(self, slot, v, fused=False):
        L, buf = self.L, self.bufs[slot]
        act = self.acts[v]
        lib = ops
        for i in range(L - 1, -1, -1):
            a, layer = act[i], self.model.layers[i]
            t_cap, t_dev = a['t_cap'], self.n_dev[slot][a['hop']]
            if i < L - 1:   # g = gradient w.r.t. relu(normalize(z)), left in dh by layer i+1
                if fused and (self.logit_fwdbwd or self.fused_pool) and i == L - 2:
                    nxt, lay2 = act[i + 1], self.model.layers[i + 1]
                    lib.l2norm_relu_bwd_z_dw(a['dh'], a['z'], True, t_cap, t_dev, a['dz'], True,
                                             nxt['dz'], self.B, self.n_dev[slot][nxt['hop']], nxt['agg'],
                                             self.gviews[lay2.lin_l.weight], a['h'] if self.root else None,
                                             self.gviews[lay2.lin_r.weight] if self.root else None,
                                             self.row_loss, self.loss)
                elif fused and self.prenorm and i == L - 2:
                    lib.l2norm_relu_bwd_z(a['dh'], a['z'], True, m=t_cap, m_dev=t_dev, dz=a['dz'], clear_dh=True)
                else:
                    lib.l2norm_relu_bwd(a['dh'], a['y'], a['inv'], True, m=t_cap, m_dev=t_dev, dz=a['dz'],
                                        clear_dh=True)   # dh is zero again for the next step's scatter-add
            x_in = self.features if i == 0 else act[i - 1]['h']
            rows = self._node_map(slot, v) if i == 0 else None
            agg_i = self._agg0(slot, v) if i == 0 else a['agg']
            if fused and i == L - 1 and self.logit_fwdbwd:
                continue      # input gradient: done by the forward launch; weight gradients: ride on layer i-1's pass
            if fused and i == L - 1 and self.fused_pool:
                # the same, and what is left of the layer is the pool Linear's own backward
                self._pool_backward(slot, i, x_in, rows, edge_pass=False)
                continue
            if fused and i == L - 1 and self.fused_logit_dw:
                # the class-logit layer's backward and its two (skinny) weight gradients in one
                # launch: the gradients' workgroups run on the CUs the 128 seed workgroups leave idle
                lib.logit_layer_bwd_dw(a['dz'], buf.seg_ptr, buf.src, self.B, self.reduce, layer.lin_l.weight,
                                       layer.lin_r.weight, act[i - 1]['dh'], agg_i, self.gviews[layer.lin_l.weight],
                                       x_in, self.gviews[layer.lin_r.weight], self.row_loss, self.loss, m_dev=t_dev)
                continue
            with self._leaf(t_cap):
                if self.root and i == 0 and self.dw_parts is not None:
                    d = self.dw_parts
                    d['n'] = lib.linear_dweight2_parts(a['dz'], agg_i, d['view_l'], x_in, d['view_r'], d['stride'], d['max'],
                                                       d['holder'], m=t_cap, m_dev=t_dev, rows2=rows)
                elif self.root:
                    lib.linear_dweight2(a['dz'], agg_i, self.gviews[layer.lin_l.weight], x_in,
                                        self.gviews[layer.lin_r.weight], m=t_cap, m_dev=t_dev, rows2=rows)
                else:
                    lib.linear_dweight(a['dz'], agg_i, m=t_cap, m_dev=t_dev, dw=self.gviews[layer.lin_l.weight])
            if self.pool:
                self._pool_backward(slot, i, x_in, rows)
            elif self.lstm:
                self._lstm_backward(slot, i)
            elif fused and i == L - 1:
                lib.logit_layer_bwd(a['dz'], buf.seg_ptr, buf.src, self.B, self.reduce, layer.lin_l.weight,
                                    layer.lin_r.weight, act[i - 1]['dh'], self.row_loss, self.loss)
            elif i > 0:
                prev = act[i - 1]
                if self.root:
                    lib.linear_dinput2(a['dz'], layer.lin_r.weight, prev['dh'], layer.lin_l.weight, a['dagg'],
                                       m=t_cap, m_dev=t_dev)
                else:      # the self-loop's share reaches prev['dh'][t] through the gcn scatter below
                    lib.linear_dinput(a['dz'], layer.lin_l.weight, m=t_cap, m_dev=t_dev, out=a['dagg'])
                lib.gather_reduce_bwd(a['dagg'], buf.seg_ptr, buf.src, t_cap, self.reduce, prev['dh'],
                                      argmax=a.get('argmax'), num_targets_dev=t_dev)
    @contextlib.contextmanager
    def _variant(self, v):
        """Variant v of the three-forward step on its stream (0: the batch itself, on the main stream)."""
        if v == 0 or self.var_streams is None:
            yield
            return
        st, main = self.var_streams[v - 1], torch.cuda.current_stream()
        if v == 1:      # (first of the forked variants: the fork point of both)
            ops.event_record(self.var_ev[0], main)
        ops.stream_wait_event(st, self.var_ev[0])
        torch.cuda.set_stream(st)
        try:
            yield
        finally:
            torch.cuda.set_stream(main)

    def _join_variants(self):
        if self.var_streams is None:
            return
        main = torch.cuda.current_stream()
        for k, st in enumerate(self.var_streams):
            ops.event_record(self.var_ev[1 + k], st)
            ops.stream_wait_event(main, self.var_ev[1 + k])

    @contextlib.contextmanager
    def _leaf(self, rows):
        """Pooling aggregators: the large weight gradients (MFMA-bound) are leaves of the backward,
        and the aggregate path that continues beside them (d agg, the edge pass into d pre:
        memory- and atomic-bound, 0.2 ms at the Reddit shape) is long enough to pay for a fork:
        1.23-1.25 against 1.26 ms per max_pool step, 1.33 against 1.355 for mean_pool
        (scratch/pool_fork_ab.sh).  In the mean / sum / max / gcn step the same fork cost more in
        cross-queue waits than the ~20 us it hid (0.41 -> 0.45 ms in round 1), so there everything
        stays on one stream."""
        if self.pool_fork is None or rows < 4096:
            yield
            return
        main = torch.cuda.current_stream()
        _lib.tape_py(lambda: self.pool_fork.wait_stream(main))
        torch.cuda.set_stream(self.pool_fork)
        try:
            yield
        finally:
            torch.cuda.set_stream(main)
        self._forked = True

    def _dagg_layer0(self, slot):
        """d agg of layer 0 = dz . W_l (the aggregate path of the pooling / lstm backward)."""
        a, layer = self.act[0], self.model.layers[0]
        t_cap, t_dev = a['t_cap'], self.n_dev[slot][a['hop']]
        if 'wlT' in a:
            wt, w = a['wlT'], layer.lin_l.weight
            _lib.tape_py(lambda: wt.copy_(w.t()))
            ops.linear_concat_fwd(a['dz'], wt, m=t_cap, m_dev=t_dev, out=a['dagg'])
        else:
            ops.linear_dinput(a['dz'], layer.lin_l.weight, m=t_cap, m_dev=t_dev, out=a['dagg'])

    def _pool_backward(self, slot, i, x_in, rows, edge_pass=True):
        """Layer i's aggregate path with a pooling aggregator: d agg -> (through the reduce) d P ->
        (relu) d pool(x) -> pool weight / bias gradients and, above layer 0, the input gradient."""
        a, layer, buf, lib = self.act[i], self.model.layers[i], self.bufs[slot], ops
        t_cap, t_dev = a['t_cap'], self.n_dev[slot][a['hop']]
        s_cap, s_dev = a['s_cap'], self.n_dev[slot][a['src_hop']]
        if not edge_pass:
            pass       # (the fused last layer: d pre is there already, the root term too)
        elif i > 0:      # root term into the previous layer's dh (overwrites its first t rows), d agg beside it
            lib.linear_dinput2(a['dz'], layer.lin_r.weight, self.act[i - 1]['dh'], layer.lin_l.weight, a['dagg'],
                               m=t_cap, m_dev=t_dev)
        else:
            self._dagg_layer0(slot)
        # the in-degree workspace (two short launches) pays for itself on the 13k targets of layer
        # 0; the B seed rows of the layers above are routed with atomics throughout
        # (max_pool 1.199 against 1.212 ms per step, mean_pool unchanged; scratch/pool_indeg_ab.sh)
        indeg = a['indeg'] if (t_cap >= 4096 or os.environ.get('SAGE_POOL_INDEG_ALL')) else None
        if not edge_pass:
            pass
        elif i == 0 and self.rev is not None:
            if self.reduce == 'max':
                lib.pool_max_bwd_pull(a['dagg'], a['arg8'], self.rev[slot], a['dpre'], num_sources_dev=s_dev)
            else:
                lib.pool_mean_bwd_pull(a['dagg'], a['P'], buf.seg_ptr, self.rev[slot], a['dpre'], num_sources_dev=s_dev)
        elif i > 0 and self.rev_up is not None and i in self.rev_up:
            lib.pool_mean_bwd_pull(a['dagg'], a['P'], buf.seg_ptr, self.rev_up[i][slot], a['dpre'], num_sources_dev=s_dev)
        elif self.reduce == 'max':
            agg_i = self.agg0[slot] if i == 0 else a['agg']
            # (the bias gradient = column sums of d pre for max as for mean: the weight-gradient
            # kernel adds them up from the rows it stages anyway, ~6 us against a 13 us pass of
            # pool_max_bwd's own over d agg)
            lib.pool_max_bwd(a['dagg'], agg_i, buf.seg_ptr, buf.src, a['argmax'], t_cap, a['dpre'],
                             num_targets_dev=t_dev, indeg=indeg)
        else:
            lib.pool_mean_bwd(a['dagg'], a['P'], buf.seg_ptr, buf.src, t_cap, a['dpre'], num_targets_dev=t_dev,
                              indeg=indeg)
        # (layer 0's is the last kernel of the backward: nothing left to run beside it)
        with self._leaf(s_cap if i > 0 else 0):
            lib.linear_dweight(a['dpre'], x_in, m=s_cap, m_dev=s_dev, rows=rows,
                               dw=self.gviews[layer.pool.weight], dbias=self.gviews[layer.pool.bias])
        if i > 0:
            # (as a forward-type bf16x3 contraction on a transposed copy of the weights, like layer 0's
            # d agg, this one is slower: it ADDS into dh — max_pool 1.172 against 1.155 ms, mean_pool
            # 1.215 against 1.198, scratch/pool_din_ab.sh)
            lib.linear_dinput(a['dpre'], layer.pool.weight, m=s_cap, m_dev=s_dev, out=self.act[i - 1]['dh'],
                              accumulate=True)

    def make_pos_neg(self, slot, counter):
        """unsup: positive = one uniform in-neighbour of every batch node in the full graph,
        negative = a uniform node (graphsage_amd/samplers/utils.py; reference samplers/utils.py:7-37).
        Only the batch's nodes (device-side count) are walked; later entries of pos_id keep older
        valid ids and are never read through a segment."""
        buf = self.bufs[slot]
        ops.random_neighbor(self.graph.colptr, self.graph.row, buf.node, self.rng_seed, counter,
                            out=self.pos_id, checked=False,
                            num_dev=buf.state[ops.ST_NODES:ops.ST_NODES + 1])
        ops.random_nodes(self.graph.num_nodes, self.pos_id.numel(), self.rng_seed, counter, out=self.neg_id)

    def all_reduce(self):
        if self.collective:
            all_reduce_flat(self.grad, self.pg, force=True)

    def optimizer_step(self, signal=None):
        """signal (0 / 1): cons_ev[signal] becomes the launch's completion event (no record() behind it)."""
        self._note('adam')
        par = self._adam_calls & 1
        self._adam_calls += 1
        if signal is None:
            self._taped('adam', par, lambda: self._optimizer_step(par))
        else:
            self._taped('adam_sig%d' % signal, par, lambda: self._optimizer_step(par, self.cons_ev[signal]))

    def _optimizer_step(self, par, done_event=None):
        if self.adam_ranges is not None:
            for k, (lo, hi) in enumerate(self.adam_ranges):
                last = k == len(self.adam_ranges) - 1
                ops.adam_step(self.flat[lo:hi], self.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                              self._steps[par:par + 1], self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                              grad_scale=1.0 / self.world_size, step_out=self._steps[1 - par:2 - par],
                              zero_grad=not self.keep_grads, done_event=done_event if last else None)
            return
        d = self.dw_parts
        parts = (d['ws'], d['stride'], d['n'], d['off'], d['len']) if (d is not None and d['n'] > 0) else None
        ops.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self._steps[par:par + 1], self.lr,
                      self.betas[0], self.betas[1], self.eps, self.weight_decay,
                      grad_scale=1.0 / self.world_size, step_out=self._steps[1 - par:2 - par],
                      zero_grad=not self.keep_grads, done_event=done_event, parts=parts)

    # ------------------------------------------------------------------ one batch at a time
    def step(self, seeds, counter=None, valid=None):
        """One eager training step on `seeds` (device int64[B]); returns the device loss tensor.
        valid < B: only the first `valid` seeds count (the partial last batch of an epoch; the
        rest of `seeds` is padding: the loss is the mean over the valid rows and the padded rows
        get a zero gradient)."""
        self._require_eager("step()")
        if self.primed:
            raise RuntimeError("a prefetched batch is pending: use pipelined_step()")
        if counter is None:
            counter = self.counter
            self.counter += 1
        self.flush()
        self._main_touched = True
        self.sample(seeds, counter, slot=0)
        if valid is not None and int(valid) < self.B:
            self._partial_forward_backward(int(valid))
        else:
            self.forward_backward(slot=0)
        self.all_reduce()
        self.optimizer_step()
        return self.loss

    def _partial_forward_backward(self, valid):
        """forward + backward of slot 0 with the loss over the first `valid` seed rows (untaped:
        once per epoch).  The padded rows' dz is zeroed so that they add nothing to any gradient."""
        if self.unsup or valid <= 0:
            raise NotImplementedError("partial batches: supervised step only")
        self.last_slot = 0
        buf, last = self.bufs[0], self.act[-1]
        self._forward_layers(0, 0, False)
        self._loss_last(0, valid)
        last['z'][valid:self.B].zero_()          # the fused kernel consumed (and cleared) `valid` rows only
        last['dz'][valid:self.B].zero_()
        self._backward(0)

    # ------------------------------------------------------------------ pipelined, eager
    def _side_gather(self, slot, before_gather, after_gather, gather_events=None):
        self.side.wait_event(self.sampled[slot])      # the sampler runs on a stream of its own
        if before_gather is None and after_gather is None:
            self.gather_layer0(slot, stream=self.side, events=gather_events)
        else:     # the hooks record events on the current stream
            with torch.cuda.stream(self.side):
                if before_gather is not None:
                    before_gather()
                self.gather_layer0(slot, events=gather_events)
                if after_gather is not None:
                    after_gather()
        self.gathered[slot] = True

    def prefetch(self, seeds, counter, before_gather=None, after_gather=None, gather_events=None, _in_step=False):
        """Sample `seeds` on the side stream into the next slot of the ring.  Call it `depth`
        times before the first pipelined_step().  (_in_step: the call comes from inside
        pipelined_step(), which has counted batch n_comp-1 as computed but not enqueued its
        compute yet — a wait on the main stream taken now covers batch n_comp-2 at most.)"""
        self._require_eager("prefetch()")
        n = self.n_pref
        if n - self.n_comp >= self.depth:
            raise RuntimeError("prefetch(): the pipeline is full")
        slot = n % self.nslots
        main = self._cur()
        need = n - self.nslots            # the batch that used this slot before
        if n < self.nslots or need > self.cons_rec_seq or self._main_touched:
            # first use of the slot, serial step()/evaluate() calls in between, or an irregular
            # call sequence that no recorded event covers
            self.samp.wait_stream(main)
            self.side_cons_seq = self.n_comp - (2 if _in_step else 1)
            self._main_touched = False
        elif need > self.side_cons_seq:
            self.samp.wait_event(self.cons_latest)      # once per group in steady state
            self.side_cons_seq = self.cons_rec_seq
        self.sample(seeds, counter, slot=slot, stream=self.samp)
        seeds.record_stream(self.samp)
        if self.overlap_gather:
            self.sampled[slot].record(self.samp)
        self.gathered[slot] = False
        self.n_pref = n + 1
        if not self.overlap_gather:
            self.ready[slot].record(self.side)
        elif self.n_pref - self.n_comp == 1:
            # nothing is computing: no GEMM to hide behind, gather right away
            self._side_gather(slot, before_gather, after_gather, gather_events)
            self.ready[slot].record(self.side)

    def pipelined_step(self, next_seeds=None, next_counter=None, before_gather=None, after_gather=None,
                       gather_events=None, infer_pred=None):
        """Train on the oldest prefetched batch while the side stream prepares the following ones:
        the sampler of batch i+depth and, with overlap_gather, the layer-0 gather of batch i+1
        (held back until this step has started, so that it runs beside the forward GEMM).
        Measurement hooks for the layer-0 gather this call launches: before_gather / after_gather
        are called around it on its stream; gather_events = (start, stop) raw HIP event handles
        are attached to the launch itself (gather_layer0).
        infer_pred (int64[B] device tensor): forward only — the same pipeline (sampler ahead, layer-0
        gather of the next batch beside this batch's contraction), no backward, no optimizer step;
        the arg-max class of every seed row is written to infer_pred (evaluate())."""
        if self.n_pref == self.n_comp:
            raise RuntimeError("pipelined_step() needs a prefetch() first")
        i = self.n_comp
        p = i % self.nslots
        ov = self.overlap_gather
        main = self._cur()
        main_first = not self._side_first
        side_next = False
        signalled = None
        if ov:
            if not self.gathered[p]:      # pipeline was drained below its depth
                self._side_gather(p, None, None)
                self.ready[p].record(self.side)
            main.wait_event(self.ready[p])
            side_next = self.n_pref - i > 1
            if side_next:
                # the gate of the next batch's gather: the main stream has reached this step.  The
                # event recorded behind the previous step's last kernel says the same (nothing but
                # the wait above sits between the two) and is already there: one marker fewer in the
                # main stream's queue per step.  SAGE_STARTED_EVENT=1: an event of its own.
                if self.cons_rec_seq == i - 1 and self.cons_latest is not None and not self._own_gate:
                    self._gate = self.cons_latest
                else:
                    self.started.record(main)
                    self._gate = self.started
                if not main_first:
                    self._gather_next(i, before_gather, after_gather, gather_events)
        else:
            if i > self.main_ready_seq:
                # one wait covers every batch prefetched so far (the side stream is in order)
                newest = self.n_pref - 1
                main.wait_event(self.ready[newest % self.nslots])
                self.main_ready_seq = newest
            if self.pool or self.lstm:
                # the pooling aggregators' "gather" starts with the pool Linear (lstm: the gates
                # contraction): it reads the weights, so the deferred Adam step of the previous
                # batch goes in front of it
                self.flush()
            if before_gather is not None:
                before_gather()
            self.gather_layer0(p, events=gather_events)
            if after_gather is not None:
                after_gather()
        self.n_comp = i + 1
        if next_seeds is not None and not main_first:
            self.prefetch(next_seeds, next_counter, _in_step=True)
        # data parallel: the previous step's all-reduce has been running on the collective's own
        # stream while this step's sampler wait and (weight-free) layer-0 gather were enqueued;
        # its Adam step goes here, just before the first kernel that reads the weights
        self.flush()
        if infer_pred is not None:
            self._infer_batch(p, infer_pred)
        else:
            self.forward_backward(slot=p, skip_gather0=True)
        if infer_pred is not None:
            pass
        elif self.collective:
            self._note('allreduce_launch')
            self._pending.launch(self.grad, self.pg, force=True, stream=self.comm)
            # bound the host's run-ahead: with RCCL neither the launch nor wait() blocks the host.
            # One event per run_ahead/2 steps, two in flight: the host is 4 to 8 steps ahead.
            half = max(1, self.run_ahead // 2)
            if (i + 1) % half == 0:
                ev = self._dp_ring[(i // half) % 3]
                ev.record(main)
                self._dp_events.append(ev)
                if len(self._dp_events) > 2:
                    self._dp_events.pop(0).synchronize()
        else:
            # the end-of-step event rides on the Adam launch as its completion event (one marker fewer
            # in the main stream's queue; SAGE_NO_ADAM_SIGNAL=1: recorded behind it as before)
            if (i + 1) % self.group == 0 and self._adam_signal:
                signalled = (i // self.group) & 1
                self.optimizer_step(signal=signalled)
            else:
                self.optimizer_step()
        if main_first:
            # the side streams' share of this call LAST: when the host is not ahead of the device
            # (the first step behind a synchronisation — a 20-step timing window starts with one —
            # or a host-bound loop), the main stream's kernels then reach the device ~0.1 ms
            # earlier; the gather is still held back by `started`, recorded in front of them.
            # (In front of this step's cons_ev record: the sampler of batch i+depth waits for the
            # compute of the batch that used its slot, not for this one.)
            if ov and side_next:
                self._gather_next(i, before_gather, after_gather, gather_events)
            if next_seeds is not None:
                self.prefetch(next_seeds, next_counter)
        if (i + 1) % self.group == 0:
            ev = self.cons_ev[(i // self.group) & 1]
            if signalled is None:
                ev.record(main)
            self.cons_latest, self.cons_rec_seq = ev, i
        return self.loss

    def _gather_next(self, i, before_gather, after_gather, gather_events):
        """overlap_gather: the layer-0 gather of batch i+1 on the side stream, released when the main
        stream has reached the start of step i (`started`)."""
        self.side.wait_event(self._gate)
        q = (i + 1) % self.nslots
        self._side_gather(q, before_gather, after_gather, gather_events)
        self.ready[q].record(self.side)

    def _infer_batch(self, slot, pred):
        """Forward of the batch in `slot` (its layer-0 aggregate is in place) down to the arg-max
        class of every seed row (reference models/graphsage.py:51-68 + the argmax of
        fig2a/runners/graphsage_runners.py:57-66).  Supervised mean / sum step: the layers below
        the last as in training, then ONE launch for the class-logit layer (sage_logit_layer_infer:
        reduce, both weight products, normalise, arg-max); recorded once per slot, replayed with
        this batch's output pointer.  Everything else: the separate kernels + torch.argmax."""
        self._note('infer')
        self.last_slot = slot
        B = self.B
        if not (pred.is_cuda and pred.dtype == torch.int64 and pred.is_contiguous()
                and pred.numel() >= B * (3 if self.multilabel else 1)):
            raise RuntimeError("infer_pred must be a contiguous CUDA int64 buffer of at least B elements "
                               "([B, 3] for multi-label targets)")
        if self.multilabel:
            # forward of every layer, then normalise + per-row (tp, fp, fn) of the prediction y > 0
            buf, last = self.bufs[slot], self.act[-1]

            def run_ml():
                self._forward_layers(slot, 0, True)
                ops.logits_bce_fused(last['z'], self.labels, buf.node, B, y=last['y'], row_stats=pred, clear_z=True)
            self._taped_last_ptr('inferml', slot, run_ml, 15, pred)
            return
        if not self.fused_logit:
            if self.unsup:
                self.make_pos_neg(slot, self.slot_counter[slot])
            self._taped('f1', slot, lambda: self._forward(slot, True))
            torch.argmax(self.act[-1]['y'][:B], dim=1, out=pred[:B])
            return
        buf, last, layer, prev = self.bufs[slot], self.act[-1], self.model.layers[-1], self.act[-2]

        def run():
            # (the normalise pass of the hidden layer stays: it is what hands z back zeroed to the
            # stream-K contraction of the next batch — or of the next TRAINING step)
            self._forward_layers(slot, 0, True, upto=self.L - 1)
            ops.logit_layer_infer(prev['h'], buf.seg_ptr, buf.src, B, self.reduce, layer.lin_l.weight,
                                  layer.lin_r.weight if self.root else layer.lin_l.weight, y=last['y'], pred=pred)
        self._taped_last_ptr('infer', slot, run, 15, pred)       # sage_logit_layer_infer(..., pred, stream)

    def _taped_last_ptr(self, what, slot, run, arg_index, out):
        """_taped() for a call sequence whose LAST C call writes to a buffer that changes from call
        to call (the evaluation pass's output slice): replayed with that one pointer patched."""
        if not self.use_tapes:
            run()
            return
        key = (what, slot, self._cur_raw())
        entry = self._tapes.get(key)
        if entry is None:
            with _lib.recording() as tape:
                run()
            fn, args, name = tape[-1]
            self._tapes[key] = (list(tape[:-1]), fn, list(args), name)
            return
        head, fn, args, name = entry
        _lib.replay(head)
        args[arg_index] = out.data_ptr()
        _lib.check(fn(*args), name)

    def flush(self):
        """Apply a deferred optimizer step.  Under data parallelism pipelined_step() returns with
        the gradient all-reduce in flight and applies Adam at the start of the next call (or
        here): call flush() before reading the weights."""
        if self._pending.active:
            self._note('allreduce_wait')
        if self._pending.complete():
            self.optimizer_step()

    # ------------------------------------------------------------------ epochs and evaluation
    def train_epoch(self, seeds, epoch=0, rank=0, drop_last=None):
        """One pass over `seeds` (device int64) in batches of B with the pipelined step.  The tail
        that does not fill a batch is trained as a padded partial batch when drop_last=False
        (the reference's loaders keep it, torch DataLoader's default; supervised step only) and
        dropped otherwise (default: under data parallelism every rank must run the same number
        of steps, which equal shards guarantee either way).  The loss is accumulated on the device — the reference
        synchronises every step with loss.item() (trainers/node_level_graphsage_trainers.py:46)
        — and returned as a device tensor: mean training loss over the epoch."""
        from .dist import batch_counter
        self._require_eager("train_epoch()")
        nb = seeds.numel() // self.B
        tail = seeds.numel() - nb * self.B if (drop_last is False and not self.unsup) else 0
        if nb == 0 and tail == 0:
            raise RuntimeError("fewer seeds than one batch")
        total = torch.zeros(1, dtype=torch.float32, device=self.dev)
        nb_all = nb + (1 if tail else 0)

        def counter(b):
            return batch_counter(epoch, nb_all, b, rank, self.world_size)
        d = self.depth
        for b in range(min(d, nb)):
            self.prefetch(seeds[b * self.B:(b + 1) * self.B], counter(b))
        for b in range(nb):
            nxt = seeds[(b + d) * self.B:(b + d + 1) * self.B] if b + d < nb else None
            loss = self.pipelined_step(nxt, counter(b + d) if nxt is not None else None)
            total.add_(loss.view(1), alpha=self.B)      # (one launch, not a multiply and an add)
        self.flush()
        if tail:
            rest = seeds[nb * self.B:]
            padded = torch.cat([rest, rest[:1].expand(self.B - tail)]).contiguous()
            total.add_(self.step(padded, counter=counter(nb), valid=tail).view(1), alpha=tail)
        # mean over the examples, as the reference accumulates it (fig2a/runners/graphsage_runners.py:48-50)
        return total / (nb * self.B + tail)

    @torch.no_grad()
    def evaluate(self, seeds, counter0=1 << 40, return_pred=False):
        """Forward-only pass over `seeds` (any length): returns (correct, count) as device int64
        tensors — accuracy = micro-F1 for single-label classification (what
        f1_score(average='micro') returns in trainers/node_level_graphsage_trainers.py:56-57).
        Sampled fan-out neighbourhoods as in GraphSAGE.inference (models/graphsage.py:51-68).
        Multi-label targets: returns (tp, fp, fn) of the prediction y > 0 over all seeds and labels
        (graph_level_graphsage_trainers.py:57-58: f1_score(y, out > 0, average='micro')).
        Runs on the training pipeline (pipelined_step(infer_pred=...)): the sampler `depth`
        batches ahead on its stream, with overlap_gather the layer-0 gather of batch b+1 beside
        the contraction of batch b, call tapes, one launch for the class-logit layer, no torch op
        per batch; the predictions are compared with the labels once, after the loop.
        return_pred: also return the predicted classes (int64[len(seeds)])."""
        self._require_eager("evaluate()")
        if self.primed:
            raise RuntimeError("a prefetched batch is pending")
        self.flush()
        B = self.B
        n = seeds.numel()
        nb = (n + B - 1) // B
        width = 3 if self.multilabel else 1
        pred_all = torch.empty(max(nb, 1) * B * width, dtype=torch.int64, device=self.dev)
        if n < nb * B:      # pad the tail with its first seed; the padded rows are not compared
            padded = torch.cat([seeds, seeds[(nb - 1) * B:(nb - 1) * B + 1].expand(nb * B - n)]).contiguous()
        else:
            padded = seeds.contiguous()
        self._main_touched = True       # (the cat above ran on the main stream: prefetch() waits for it)
        d = self.depth
        # Pooling aggregators: relu(pool(x)) is row-wise and the weights stand still during the
        # pass, so when the batches together would push more rows through layer 0's pool Linear
        # than the graph has nodes (Reddit's test split: 109 batches x 81 k rows against 233 k
        # nodes) it is computed ONCE over the whole feature table and every batch gathers from
        # that — the same rows through the same kernel, 1.1 ms instead of 109 x 0.38.
        n_nodes = self.features.size(0)
        if self.pool and not os.environ.get('SAGE_NO_EVAL_POOL_ALL') and nb * min(self.act[0]['s_cap'], n_nodes) > 2 * n_nodes:
            layer = self.model.layers[0]
            if self._p_all_buf is None:
                fin = self.act[0]['fin']
                self._p_all_buf = torch.empty((n_nodes, (fin + 15) // 16 * 16), dtype=torch.float32, device=self.dev)[:, :fin]
            ops.pool_linear_fwd(self.features, layer.pool.weight, layer.pool.bias, m=n_nodes, out=self._p_all_buf)
            self._p_all = self._p_all_buf
        self._eval_pass = True
        try:
            for b in range(min(d, nb)):
                self.prefetch(padded[b * B:(b + 1) * B], counter0 + b)
            for b in range(nb):
                nxt = padded[(b + d) * B:(b + d + 1) * B] if b + d < nb else None
                self.pipelined_step(nxt, counter0 + b + d if nxt is not None else None,
                                    infer_pred=pred_all[b * B * width:(b + 1) * B * width])
        finally:
            self._p_all = None
            self._eval_pass = False
        if self.multilabel:      # (tp, fp, fn) over the real seeds: micro-F1 = 2 tp / (2 tp + fp + fn)
            tp, fp, fn = pred_all.view(-1, 3)[:n].sum(dim=0).view(3, 1).unbind(0)
            return tp, fp, fn
        correct = (pred_all[:n] == self.labels[seeds]).sum().view(1)
        count = torch.full((1,), n, dtype=torch.int64, device=self.dev)
        if return_pred:
            return correct, count, pred_all[:n]
        return correct, count

    def evaluate_loader(self, loader, return_pred=False):
        """evaluate() on the seeds of an evaluation loader (graphsage_amd.samplers.UniformLoader),
        with the Philox counters one pass over that loader would use — the sampled neighbourhoods
        are the ones GraphSAGE.inference(loader) would see (models/graphsage.py:51-68) — and the
        loader's epoch advanced by the TWO passes the reference makes per evaluation (inference,
        then a second walk for the labels: trainers/node_level_graphsage_trainers.py:54-55,
        fig2a/runners/graphsage_runners.py:60-62).  None when the loader's batches are not the
        engine's (other batch size / fan-out / sampler seed / graph, shuffled or sharded seeds):
        the caller then walks the loader itself."""
        from .dist import batch_counter
        samp = loader.neighbor_sampler
        if self.multilabel and return_pred:
            return None
        if (loader.batch_size != self.B or list(samp.num_neighbors) != self.fanout or samp.seed != self.rng_seed
                or samp.graph is not self.graph or loader.shuffle or loader.world_size != 1 or loader.drop_last
                or loader.transform is not None or len(loader.dataset) == 0):
            return None
        counter0 = batch_counter(loader.epoch, len(loader), 0, 0, 1)
        out = self.evaluate(loader._idx_dev, counter0=counter0, return_pred=return_pred)
        loader.epoch += 2
        return out

    # ------------------------------------------------------------------ pipelined, HIP graphs
    def capture(self, seed_pool, counter0=0, counter_stride=1, split_gather=False):
        """Capture the pipelined step into HIP graphs.  Replay b uses seeds
        seed_pool[b*B:(b+1)*B] (wrapping at the end of the pool) and Philox counter
        counter0 + b*counter_stride, exactly what step(seed_pool[...], counter) would use.

        Per slot: graph S (sampler + control-block advance, replayed on the side stream) and
        graph C (forward/backward); one graph A (Adam).  The RCCL all-reduce stays eager between
        C and A.  split_gather leaves the layer-0 gather outside graph C so a caller can bracket
        that one kernel with events (bench.py's roofline probe)."""
        if seed_pool.numel() < self.B:
            raise RuntimeError("seed pool smaller than one batch")
        if self.primed:
            raise RuntimeError("a prefetched batch is pending")
        if self.overlap_gather or self.unsup:
            raise NotImplementedError("overlap_gather / unsup are options of the eager pipeline")
        self.pool_fork = None      # (the captured step stays on one stream)
        self._tapes.clear()        # eager tapes hold calls bound to the fork stream
        self.flush()
        self.seed_pool = seed_pool.contiguous()
        self.split_gather = split_gather
        pool_len = (self.seed_pool.numel() // self.B) * self.B
        self.ctrl.zero_()
        self.ctrl[ops.CTRL_STAMP] = self._stamp    # stamps already used eagerly stay below
        torch.cuda.synchronize()
        cap = torch.cuda.Stream(device=self.dev)
        cap.wait_stream(torch.cuda.current_stream())
        gs = [torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()]
        gc = [torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()]
        ga = [torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()]   # one per step-counter parity
        with torch.cuda.stream(cap):
            for slot in (0, 1):
                with torch.cuda.graph(gs[slot], stream=cap):
                    self.sample(self.seed_pool, counter0, slot=slot, ctrl=self.ctrl)
                    ops.ctrl_advance(self.ctrl, counter_stride, 1, self.B, pool_len)
                with torch.cuda.graph(gc[slot], stream=cap):
                    self.forward_backward(slot=slot, skip_gather0=split_gather)
            # capture executes nothing: the host-side parity must end where it started, and the
            # first replay must read the word the eager steps wrote last
            par0 = self._adam_calls & 1
            for par in (par0, 1 - par0):
                with torch.cuda.graph(ga[par], stream=cap):
                    self._optimizer_step(par)
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        self.graphs = (gs, gc, ga)
        self.use = 0
        # prime: sample the first batch
        with torch.cuda.stream(self.side):
            self.side.wait_stream(torch.cuda.current_stream())
            gs[0].replay()
            self.ready[0].record(self.side)
        self._first_replay = True
        return self

    def replay(self, before_gather=None, after_gather=None, gather_events=None):
        gs, gc, ga = self.graphs
        p = self.use
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self.side):            # next batch into the other slot
            if not self._first_replay:
                self.side.wait_event(self.consumed[1 - p])
            gs[1 - p].replay()
            self.ready[1 - p].record(self.side)
        self._first_replay = False
        main.wait_event(self.ready[p])
        if self.split_gather:
            if before_gather is not None:
                before_gather()
            self.gather_layer0(p, events=gather_events)
            if after_gather is not None:
                after_gather()
        gc[p].replay()
        self.last_slot = p
        self.all_reduce()
        ga[self._adam_calls & 1].replay()
        self._adam_calls += 1
        self.consumed[p].record(main)
        self.use = 1 - p
        return self.loss
