"""GPU parity: the static-shape engine (what bench.py times) computes the same training
steps as the module path (GraphSAGE.forward_batch + F.cross_entropy + autograd +
torch.optim.Adam), which test_gpu_model.py ties to the CPU oracle."""
import copy

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import assert_trajectories_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("aggregator", ["mean", "sum", "max", "gcn", "max_pool", "mean_pool", "lstm",
                                        "max_pool/pull", "mean_pool/pull"])
def test_engine_matches_module_path(dev, aggregator, monkeypatch):
    from graphsage_amd import settings
    from graphsage_amd.datasets import planted_partition
    monkeypatch.setattr(settings, "LSTM_NUM_INPUTS", 4)      # fan-outs 5 / 3: truncation and padding both occur
    # "/pull": layer 0's edge pass pulled by the source rows (what Reddit-sized batches take by
    # themselves: engine.rev, sage_pool_reverse_edges behind the sampler)
    aggregator, _, pull = aggregator.partition("/")
    if pull:
        monkeypatch.setenv("SAGE_POOL_PULL_ALL", "1")
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import NeighborSampler, UniformLoader
    from graphsage_amd.samplers.uniform_loader import device_graph
    data = planted_partition(3000, 60000, 40, 6, seed=2)
    B = 64
    loader = UniformLoader(data, num_neighbors=[5, 3], input_nodes=torch.arange(4 * B), batch_size=B, seed=7)
    torch.manual_seed(0)
    model_a = GraphSAGE(40, 32, aggregator, 2, 6).to(dev)
    model_b = copy.deepcopy(model_a)
    graph = device_graph(data)
    engine = SageTrainEngine(model_b, NeighborSampler(data, [5, 3], seed=7), graph.features(data),
                             graph.labels(data), B, lr=1e-2, keep_grads=True)
    assert (engine.rev is not None) == bool(pull)
    opt = torch.optim.Adam(model_a.parameters(), lr=1e-2)
    for b, batch in enumerate(loader):
        opt.zero_grad()
        loss_a = F.cross_entropy(model_a.forward_batch(batch), batch.y[:B])
        loss_a.backward()
        # (lstm: W_hh multiplies the zero initial state — no gradient on the module path, zeros here)
        grads_a = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in model_a.parameters()]
        opt.step()
        seeds = torch.arange(b * B, (b + 1) * B, device=dev)
        loss_b = engine.step(seeds, counter=b)
        torch.testing.assert_close(loss_b[0], loss_a.detach(), rtol=1e-5, atol=1e-6)
        for ga, pb in zip(grads_a, model_b.parameters()):
            torch.testing.assert_close(engine.gviews[pb], ga, rtol=1e-3, atol=1e-6)
        for pa, pb in zip(model_a.parameters(), model_b.parameters()):
            if aggregator == "lstm":
                # 25,600 cell weights, some with gradients near Adam's eps: the first steps move such
                # an element by up to lr whatever the gradient's size, so a last-bit difference shows
                off = (pb.data - pa.data).abs() > 1e-5 + 1e-4 * pa.data.abs()
                assert int(off.sum()) <= max(1, off.numel() // 1000), int(off.sum())
                pa.data.copy_(pb.data)
            else:
                torch.testing.assert_close(pb.data, pa.data, rtol=1e-4, atol=1e-5)
    assert int(engine.step_dev) == 4


@pytest.mark.parametrize("aggregator,keep", [("mean", True), ("mean", False), ("sum", False), ("max", False), ("gcn", True),
                                             ("max_pool", True), ("mean_pool", True)])
def test_three_layer_engine_matches_module_path(dev, aggregator, keep, monkeypatch):
    """Three SAGE layers (reference models/graphsage.py:44-49 with num_layers = 3): the class-logit
    layer's one-launch forward + backward sits on layer 2, its riders on layer 1's pass, the stored
    weight-gradient slices on layer 0 (keep=False) — losses and, with keep_grads, every gradient
    against the module path; weights after Adam either way.  The pooling aggregators: their one-launch
    last layer on layer 2, the riders on layer 1's pass, the pulled edge pass on layer 0 and (mean_pool)
    layer 1."""
    if aggregator.endswith("_pool"):
        monkeypatch.setenv("SAGE_POOL_PULL_ALL", "1")
    from graphsage_amd.datasets import planted_partition
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import NeighborSampler, UniformLoader
    from graphsage_amd.samplers.uniform_loader import device_graph
    data = planted_partition(2500, 40000, 40, 6, seed=6)
    B, fan = 48, [5, 3, 2]
    loader = UniformLoader(data, num_neighbors=fan, input_nodes=torch.arange(4 * B), batch_size=B, seed=5)
    torch.manual_seed(4)
    model_a = GraphSAGE(40, 32, aggregator, 3, 6).to(dev)
    model_b = copy.deepcopy(model_a)
    graph = device_graph(data)
    engine = SageTrainEngine(model_b, NeighborSampler(data, fan, seed=5), graph.features(data), graph.labels(data), B,
                             lr=1e-2, keep_grads=keep)
    if aggregator.endswith("_pool"):
        assert engine.fused_pool and engine.rev is not None and (engine.rev_up is not None) == (aggregator == "mean_pool")
        assert engine.rev_up is None or sorted(engine.rev_up) == [1]
    else:
        assert engine.logit_fwdbwd and (engine.dw_parts is None) == keep          # (max as well: its arg-max stays in the wave)
    opt = torch.optim.Adam(model_a.parameters(), lr=1e-2)
    for b, batch in enumerate(loader):
        opt.zero_grad()
        loss_a = F.cross_entropy(model_a.forward_batch(batch), batch.y[:B])
        loss_a.backward()
        opt.step()
        loss_b = engine.step(torch.arange(b * B, (b + 1) * B, device=dev), counter=b)
        torch.testing.assert_close(loss_b[0], loss_a.detach(), rtol=1e-5, atol=1e-6)
        if keep:
            for pa, pb in zip(model_a.parameters(), model_b.parameters()):
                torch.testing.assert_close(engine.gviews[pb], pa.grad, rtol=1e-3, atol=1e-6)
            engine.grad.zero_()
        for pa, pb in zip(model_a.parameters(), model_b.parameters()):
            torch.testing.assert_close(pb.data, pa.data, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("fanout,n_in,padsum", [([4, 2, 3], 6, True), ([6, 4], 4, False), ([3, 5], 4, False)])
def test_lstm_engine_padded_slot_split(dev, fanout, n_in, padsum, monkeypatch):
    """lstm layer 0 on the engine: the rows behind the seeds contract their padded slots once
    (engine.make_act `kpad`, csrc/lstm.hip `lstm_pad_weight_sum_kernel`) — three hops (kpad = the
    largest of the later fan-outs), and the cases that must NOT take the split (later fan-outs at or
    above LSTM_NUM_INPUTS) — each against the module path's
    gradients (reference layers/sage.py:147-193 as read in layers/lstm.py)."""
    from graphsage_amd import settings
    from graphsage_amd.datasets import planted_partition
    monkeypatch.setattr(settings, "LSTM_NUM_INPUTS", n_in)
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import NeighborSampler, UniformLoader
    from graphsage_amd.samplers.uniform_loader import device_graph
    data = planted_partition(1500, 18000, 24, 5, seed=11)
    B, L = 32, len(fanout)
    loader = UniformLoader(data, num_neighbors=fanout, input_nodes=torch.arange(3 * B), batch_size=B, seed=3)
    torch.manual_seed(1)
    model_a = GraphSAGE(24, 16, "lstm", L, 5).to(dev)
    model_b = copy.deepcopy(model_a)
    graph = device_graph(data)
    engine = SageTrainEngine(model_b, NeighborSampler(data, fanout, seed=3), graph.features(data),
                             graph.labels(data), B, lr=1e-2, keep_grads=True)
    assert ('kpad' in engine.act[0]) == padsum
    if padsum:
        assert engine.act[0]['kpad'] == max(fanout[1:]) and all('kpad' not in a for a in engine.act[1:])
    for b, batch in enumerate(loader):
        model_a.zero_grad()
        loss_a = F.cross_entropy(model_a.forward_batch(batch), batch.y[:B])
        loss_a.backward()
        loss_b = engine.step(torch.arange(b * B, (b + 1) * B, device=dev), counter=b)
        torch.testing.assert_close(loss_b[0], loss_a.detach(), rtol=1e-5, atol=1e-6)
        for pa, pb in zip(model_a.parameters(), model_b.parameters()):
            ga = pa.grad if pa.grad is not None else torch.zeros_like(pa)
            torch.testing.assert_close(engine.gviews[pb], ga, rtol=1e-3, atol=1e-6)
            pa.data.copy_(pb.data)          # same weights for the next batch (Adam ran on the engine only)


@pytest.mark.parametrize("split,aggregator", [(False, "mean"), (True, "mean"), (False, "max_pool"), (False, "mean_pool")])
def test_graph_replay_matches_eager_steps(dev, split, aggregator, monkeypatch, overlap=False):
    """HIP-graph replay (device control block drives seeds / Philox counter / stamp) runs the
    same steps as eager engine.step, including the wrap at the end of the seed pool.  Pooling
    aggregators: with the per-source edge lists (scan included) captured behind the sampler."""
    if aggregator.endswith("_pool"):
        monkeypatch.setenv("SAGE_POOL_PULL_ALL", "1")
    from graphsage_amd.datasets import planted_partition
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import NeighborSampler
    from graphsage_amd.samplers.uniform_loader import device_graph
    data = planted_partition(2000, 40000, 24, 5, seed=4)
    B = 32
    graph = device_graph(data)
    torch.manual_seed(1)
    model_a = GraphSAGE(24, 16, aggregator, 2, 5).to(dev)
    model_b = copy.deepcopy(model_a)
    mk = lambda m, ov=False: SageTrainEngine(m, NeighborSampler(data, [4, 3], seed=9), graph.features(data),
                                             graph.labels(data), B, lr=1e-2, overlap_gather=ov)
    eager, graphed = mk(model_a), mk(model_b, overlap)
    assert (graphed.rev is not None) == aggregator.endswith("_pool")
    pool = torch.randperm(2000, generator=torch.Generator().manual_seed(0))[:3 * B + 7].to(dev)  # 3 batches
    graphed.step(pool[:B], counter=1000)      # an eager step before capture (stamps already in use)
    eager.step(pool[:B], counter=1000)
    graphed.capture(pool, counter0=5, counter_stride=2, split_gather=split)
    for s in range(7):                         # wraps twice
        b = s % 3
        la = eager.step(pool[b * B:(b + 1) * B], counter=5 + 2 * s).clone()
        lb = graphed.replay().clone()
        assert torch.equal(eager.buf.node[:B], graphed.buf.node[:B])
        assert torch.equal(eager.buf.state, graphed.buf.state)
        assert eager.buf is eager.bufs[0] and graphed.buf is graphed.bufs[s % 2]
        n, e = int(eager.buf.state[0]), int(eager.buf.state[1])
        assert torch.equal(eager.buf.node[:n], graphed.buf.node[:n])
        assert torch.equal(eager.buf.src[:e], graphed.buf.src[:e])
        torch.testing.assert_close(lb, la, rtol=1e-5, atol=1e-6)
        assert_trajectories_close(graphed.flat, eager.flat)
    assert int(graphed.state_sum[4]) == 0 and int(graphed.step_dev) == 8
    with pytest.raises(RuntimeError):
        graphed.step(pool[:B])


@pytest.mark.parametrize("overlap", [False, True])
def test_pipelined_eager_matches_serial_steps(dev, overlap):
    """prefetch()/pipelined_step(): the sampler of step s+1 runs on a side stream during step s;
    the batches and the training trajectory are those of serial step() calls."""
    from graphsage_amd.datasets import planted_partition
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import NeighborSampler
    from graphsage_amd.samplers.uniform_loader import device_graph
    data = planted_partition(2000, 40000, 24, 5, seed=5)
    B = 32
    graph = device_graph(data)
    torch.manual_seed(2)
    model_a = GraphSAGE(24, 16, "mean", 2, 5).to(dev)
    model_b = copy.deepcopy(model_a)
    mk = lambda m, ov=False: SageTrainEngine(m, NeighborSampler(data, [4, 3], seed=9), graph.features(data),
                                             graph.labels(data), B, lr=1e-2, overlap_gather=ov)
    serial, piped = mk(model_a), mk(model_b, overlap)
    pool = torch.randperm(2000, generator=torch.Generator().manual_seed(1))[:20 * B].to(dev)
    d = piped.depth
    assert d == (2 if overlap else 4)
    for s in range(d):
        piped.prefetch(pool[s * B:(s + 1) * B], s)
    with pytest.raises(RuntimeError):
        piped.prefetch(pool[:B], 99)              # pipeline full
    for s in range(20):                          # the slot ring (8 or 3 slots) wraps
        la = serial.step(pool[s * B:(s + 1) * B], counter=s).clone()
        nxt = pool[(s + d) * B:(s + d + 1) * B] if s + d < 20 else None
        lb = piped.pipelined_step(nxt, s + d).clone()
        n, e = int(serial.buf.state[0]), int(serial.buf.state[1])
        assert torch.equal(serial.buf.state, piped.buf.state)
        assert torch.equal(serial.buf.node[:n], piped.buf.node[:n])
        assert torch.equal(serial.buf.src[:e], piped.buf.src[:e])
        torch.testing.assert_close(lb, la, rtol=1e-5, atol=1e-6)
        assert_trajectories_close(piped.flat, serial.flat)
    assert not piped.primed
    with pytest.raises(RuntimeError):
        piped.pipelined_step()
    # drained below its depth and refilled with ONE batch: still correct (gather on demand)
    piped.prefetch(pool[:B], 50)
    la = serial.step(pool[:B], counter=50).clone()
    lb = piped.pipelined_step().clone()
    torch.testing.assert_close(lb, la, rtol=1e-5, atol=1e-6)
    assert torch.equal(serial.state_sum, piped.state_sum)
    piped.step(pool[:B], counter=99)          # pipeline drained: serial stepping works again


@pytest.mark.parametrize("aggregator,overlap", [("mean", False), ("mean", True), ("sum", True), ("max", False),
                                                ("max", True), ("gcn", True), ("max_pool", False), ("mean_pool", False),
                                                ("lstm", False)])
def test_evaluate_pipeline_matches_module_inference(dev, aggregator, overlap, monkeypatch):
    """engine.evaluate() — the training pipeline run forward-only, one launch for the class-logit
    layer of the mean / sum step — predicts what GraphSAGE.inference + argmax predict
    (reference models/graphsage.py:51-68, fig2a/runners/graphsage_runners.py:57-66) on the same
    sampled batches; the tail that does not fill a batch included, and a training step afterwards
    still finds its accumulation buffers zeroed.  Pooling aggregators: six batches of up to
    1,344 sampled nodes against 3,000 nodes in the graph — the pass computes relu(pool(x)) once
    over all nodes and the batches gather from it (engine._p_all), the reference side runs the
    pool Linear per batch."""
    from graphsage_amd import settings
    from graphsage_amd.datasets import planted_partition
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import NeighborSampler, UniformLoader
    from graphsage_amd.samplers.uniform_loader import device_graph
    monkeypatch.setattr(settings, "LSTM_NUM_INPUTS", 4)
    data = planted_partition(3000, 60000, 40, 6, seed=2)
    B = 64
    n_eval = 5 * B + 17
    ids = torch.randperm(3000, generator=torch.Generator().manual_seed(3))[:n_eval]
    torch.manual_seed(0)
    model = GraphSAGE(40, 32, aggregator, 2, 6).to(dev)
    graph = device_graph(data)
    engine = SageTrainEngine(copy.deepcopy(model), NeighborSampler(data, [5, 3], seed=7), graph.features(data),
                             graph.labels(data), B, lr=1e-2, overlap_gather=overlap)
    ref_engine = SageTrainEngine(copy.deepcopy(model), NeighborSampler(data, [5, 3], seed=7), graph.features(data),
                                 graph.labels(data), B, lr=1e-2)
    # a few training steps first: evaluate() must cope with a used pipeline and leave it usable
    train = torch.arange(4 * B, device=dev)
    for eng in (engine, ref_engine):
        eng.train_epoch(train, epoch=0)
    for _ in range(2):
        correct, count, pred = engine.evaluate(ids.to(dev), counter0=900, return_pred=True)
        assert int(count) == n_eval and pred.numel() == n_eval
        assert (engine._p_all_buf is not None) == aggregator.endswith("_pool") and engine._p_all is None
        # the same batches through the separate kernels of forward() + torch.argmax
        want, margin = [], []
        padded = torch.cat([ids, ids[5 * B:5 * B + 1].expand(6 * B - n_eval)]).to(dev)
        for b in range(6):
            ref_engine.sample(padded[b * B:(b + 1) * B].contiguous(), 900 + b, slot=0)
            ref_engine.forward(slot=0)
            y = ref_engine.act[-1]['y'][:B]
            top = y.topk(2, dim=1).values
            want.append(y.argmax(dim=1))
            margin.append(top[:, 0] - top[:, 1])
        want, margin = torch.cat(want)[:n_eval], torch.cat(margin)[:n_eval]
        differ = pred != want
        assert not bool((differ & (margin > 1e-5)).any())      # only numerical ties may flip
        assert int(differ.sum()) <= 2
        assert int(correct) == int((pred == graph.labels(data)[ids.to(dev)]).sum())
        # ... and training goes on as if nothing had happened in between
        for eng in (engine, ref_engine):
            eng.train_epoch(train, epoch=1)
        if aggregator == "lstm":
            # (split-K partial sums arrive in any order; Adam turns a last-bit difference of a gradient
            # near its eps into a visible one, see test_engine_matches_module_path)
            off = (engine.flat - ref_engine.flat).abs() > 1e-6 + 1e-4 * ref_engine.flat.abs()
            assert int(off.sum()) <= max(1, off.numel() // 1000), int(off.sum())
            ref_engine.flat.copy_(engine.flat)
        else:
            assert_trajectories_close(engine.flat, ref_engine.flat)


def test_event_record_and_stream_wait_order_two_streams(dev):
    """sage_event_record / sage_stream_wait_event (the fork and join of the forked variants): work
    enqueued on stream b behind the wait sees what stream a wrote in front of the record, both as
    direct calls and replayed from a recorded tape."""
    from graphsage_amd import _lib, ops
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    ev = ops.new_event()
    x = torch.zeros(1 << 20, device=dev)
    y = torch.zeros(1 << 20, device=dev)
    torch.cuda.synchronize()

    def round_trip(tape=None):
        with torch.cuda.stream(a):
            torch.cuda._sleep(5_000_000)           # a few ms: b would run ahead without the wait
            x.add_(1.0)
        if tape is None:
            ops.event_record(ev, a)
            ops.stream_wait_event(b, ev)
        else:
            _lib.replay(tape)
        with torch.cuda.stream(b):
            y.copy_(x * 2.0)
        torch.cuda.synchronize()

    with _lib.recording() as tape:
        round_trip()
    assert len(tape) == 2 and float(y.min()) == 2.0 and float(y.max()) == 2.0
    round_trip(tape)
    assert float(y.min()) == 4.0 and float(y.max()) == 4.0
    lib = _lib.load()
    assert lib.sage_event_record(None, None) == -1 and "NULL event" in _lib.last_error()
    assert lib.sage_stream_wait_event(None, None) == -1


@pytest.mark.parametrize("aggregator,unsup", [("mean", False), ("max", False), ("gcn", False), ("max_pool", False),
                                              ("mean_pool", False), ("mean", True), ("mean", "only"), ("max", "only")])
def test_engine_hands_its_accumulation_buffers_back_zeroed(dev, aggregator, unsup):
    """The promise behind SAGE_EPI_ZEROED and behind the scatter-adds of the backward: after a
    step — and after an evaluation pass — every pre-activation buffer z and every hidden-gradient
    buffer dh of every variant holds zeros again (their consumers clear what they read)."""
    from graphsage_amd.datasets import planted_partition
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import NeighborSampler
    from graphsage_amd.samplers.uniform_loader import device_graph
    data = planted_partition(3000, 60000, 40, 6, seed=2)
    B = 64
    torch.manual_seed(0)
    model = GraphSAGE(40, 32, aggregator, 3, 6).to(dev)
    graph = device_graph(data)
    engine = SageTrainEngine(model, NeighborSampler(data, [5, 3, 2], seed=7), graph.features(data),
                             graph.labels(data), B, lr=1e-2, unsup=unsup)
    assert engine.zeroed_hint

    def check(when):
        for v, act in enumerate(engine.acts):
            for i, a in enumerate(act):
                for name in ("z", "dh"):
                    t = a.get(name)
                    if t is not None:
                        assert not bool(t.any()), (when, "variant", v, "layer", i, name)

    for b in range(3):
        engine.step(torch.arange(b * B, (b + 1) * B, device=dev), counter=b)
        torch.cuda.synchronize()
        check("after step %d" % b)
    if not unsup:
        engine.evaluate(torch.arange(5 * B + 17, device=dev))
        torch.cuda.synchronize()
        check("after evaluate")
        engine.step(torch.arange(B, device=dev), counter=9)
        torch.cuda.synchronize()
        check("after the step behind evaluate")
