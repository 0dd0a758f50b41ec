#!/usr/bin/env python3
"""bench.py — Reddit-shaped GraphSAGE-mean 25/10 training step throughput on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched
one rank per GPU by torch.distributed.run.  Prints ONE JSON line on rank 0.

A step is fig2a's (reference experiments/fig2a/runners/graphsage_runners.py:38-46): sample
(25,10) -> gather -> 2-layer GraphSAGE-mean forward -> cross-entropy on the 512 seed rows ->
backward -> Adam; every rank does its own 512 seeds per step (weak scaling), gradients are
all-reduced over RCCL.  Inputs (graph, features, labels, weights) are resident in HBM before
the timed region.  `roofline` is for the dominant HBM-bound kernel, the layer-0 fused
gather+mean (see DESIGN.md for the byte formula); `cpu_baseline` times the oracle's CPU
restatement of the same step (rank 0, N = 1 only) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

SHAPES = {
    # name: (num_nodes, num_edges, num_features, num_classes, num_train)
    "reddit": (232965, 114615892, 602, 41, 153431),
    "reddit-small": (23296, 4000000, 602, 41, 15343),   # for quick functional runs only
    # BASELINE.json configs[4] (ogbn-products scale; use with --variant unsup): informational
    "products": (2400000, 62000000, 100, 47, 196615),
    # BASELINE.json configs[1] (PPI's size as ONE graph: 56,944 nodes, 818,716 edges, 50 features, 121
    # independent sigmoid labels, BCE-with-logits loss; 44,906 training nodes): informational
    "ppi": (56944, 818716, 50, 121, 44906),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--settle-steps", type=int, default=1500,
                    help="untimed steps BEFORE the warm-up steps (set-up, not part of W or K): the first "
                         "~0.5 s of launches of a fresh process run at half speed on the host side "
                         "(lazy code-object loading, call-tape recording, and whatever the driver still "
                         "does for a process that has just exited)")
    ap.add_argument("--workload", default="reddit", choices=sorted(SHAPES))
    ap.add_argument("--batch-size", type=int, default=512)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--fanout", type=int, nargs="+", default=[25, 10])
    ap.add_argument("--aggregator", default="mean")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--cpu-steps", type=int, default=20,
                    help="timed CPU-baseline steps (0 = skip); SURVEY 8(d): at least 20 after 5 warm-ups, median")
    ap.add_argument("--cpu-warmup", type=int, default=5, help="untimed CPU-baseline steps in front of the timed ones")
    ap.add_argument("--group", type=int, default=4,
                    help="batches per cross-stream synchronisation of the eager pipeline (engine `group`)")
    ap.add_argument("--no-variant", action="store_true",
                    help="skip the informational extra measurement (exact-fp32 MFMA mode of the GEMMs)")
    ap.add_argument("--no-probe", action="store_true", help="do not time the gather kernel with events")
    ap.add_argument("--bracket-probe", action="store_true",
                    help="record the probe's events around the gather launch instead of attaching them to it")
    ap.add_argument("--serial", action="store_true",
                    help="headline from the serial schedule (every kernel of a step on one stream) instead of "
                         "the overlapped one (layer-0 gather of batch s+1 on the side stream beside batch s)")
    ap.add_argument("--overlapped", action="store_true",
                    help="the overlapped schedule also for rows too narrow for the engine's overlap_gather='auto' "
                         "rule (an A/B switch; one GPU, not the pooling / lstm aggregators)")
    ap.add_argument("--module-path", action="store_true",
                    help="max_pool / mean_pool through the module path (autograd Functions) instead of the engine")
    ap.add_argument("--probe-steps", type=int, default=300,
                    help="timed steps of the serial roofline probe run (gather kernel alone on its stream)")
    ap.add_argument("--lstm-num-inputs", type=int, default=25, help="--aggregator lstm: LSTM_NUM_INPUTS")
    ap.add_argument("--variant", default="fig2a", choices=["fig2a", "trainer", "trainer-module", "unsup"],
                    help="fig2a: the benchmarked step (supervised term only). trainer: the reference "
                         "trainers' step (node_level_graphsage_trainers.py:34-46: + unsupervised term = two "
                         "more forwards on positive / negative nodes) on the static-shape engine; "
                         "trainer-module: the same through the module path (autograd Functions).  Both "
                         "are reported separately, never as the headline.  unsup: the unsupervised "
                         "trainer's step (:104-117), loss = the random-walk term alone, embedding width = "
                         "--hidden (BASELINE.json configs[4] with --workload products)")
    ap.add_argument("--force-collective", action="store_true",
                    help="one GPU: run the data-parallel control flow anyway (a 1-rank RCCL group, asynchronous "
                         "all-reduce on RCCL's stream, deferred Adam, serial schedule) — measures what that control "
                         "flow costs per step before any link latency; informational, never the driver's command")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from HIP graphs instead of launching every kernel eagerly "
                         "(same arithmetic; measured 2-3 %% slower than eager on MI355X / ROCm 7.2)")
    ap.add_argument("--no-graph", action="store_true", help="(default now; kept for old command lines)")
    return ap.parse_args()


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(data_cpu, colptr, row, seeds_list, fanout, layers, steps, lr, warmup=5):
    """The oracle's restatement of the reference step, as written (unfused, all sampled rows
    through both layers), on the host cores: `warmup` untimed steps, then `steps` timed ones.
    Returns (seeds/s from the MEDIAN step time, cores, median s/step, timed steps, mean s/step)."""
    import numpy as np
    import torch.nn.functional as F

    from oracle import sage_ref, sampler, sampler_c
    cores = os.cpu_count() or 1
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    torch.set_num_threads(cores)
    x, y = data_cpu
    params = []
    ref_layers = []
    for p in layers:
        d = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        ref_layers.append(d)
        params += list(d.values())
    opt = torch.optim.Adam(params, lr=lr)
    samp = sampler_c.neighbor_sample if sampler_c.available() else sampler.neighbor_sample
    times = []
    for i, seeds in enumerate(seeds_list[: steps + warmup]):
        t0 = time.perf_counter()
        node, src, dst, eid, _, _ = samp(colptr, row, seeds, fanout, 0, i)
        opt.zero_grad()
        loss, _ = sage_ref.train_step_loss(x, y, torch.from_numpy(np.asarray(node)), torch.from_numpy(np.asarray(src)),
                                           torch.from_numpy(np.asarray(dst)), len(seeds), ref_layers)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    timed = times[warmup:] if len(times) > warmup else times[-1:]
    per_step = sorted(timed)[len(timed) // 2] if len(timed) % 2 else \
        0.5 * (sorted(timed)[len(timed) // 2 - 1] + sorted(timed)[len(timed) // 2])
    return len(seeds_list[0]) / per_step, cores, per_step, len(timed), sum(timed) / len(timed)


def trainer_variant(args, data, graph, model, N, E, Fdim, C, n_train, B, K, world, rank):
    """The reference trainers' step (3 forwards) through the module path (autograd Functions over
    the same kernels): sample -> forward -> CE + unsupervised loss (positive = random in-neighbour,
    negative = random node, both forwarded over the batch's edge_index) -> backward -> Adam."""
    import torch.nn.functional as F

    from graphsage_amd.samplers import UniformLoader, get_pos_neg_batches
    if world != 1:
        raise SystemExit("--variant trainer is a single-GPU measurement")
    gen = torch.Generator().manual_seed(0)
    train = torch.randperm(N, generator=gen)[:n_train]
    loader = UniformLoader(data, num_neighbors=K, input_nodes=train, batch_size=B, shuffle=False, seed=0)
    opt = torch.optim.Adam(model.parameters(), lr=args.lr)

    def step(batch):
        opt.zero_grad()
        y_hat = model.forward_batch(batch)
        pos, neg = get_pos_neg_batches(batch, data)
        pos_out = model(pos.x, pos.edge_index)[:batch.batch_size]
        neg_out = model(neg.x, neg.edge_index)[:batch.batch_size]
        unsup = -F.logsigmoid((y_hat * pos_out).sum(-1)).mean() - F.logsigmoid(-(y_hat * neg_out).sum(-1)).mean()
        loss = F.cross_entropy(y_hat, batch.y[:batch.batch_size]) + unsup
        loss.backward()
        opt.step()
        return loss

    it = iter(loader)
    for _ in range(args.warmup):
        step(next(it))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(next(it))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    print(json.dumps({
        "metric": "Reddit GraphSAGE-mean 25/10 fwd+bwd seed nodes/s; gather GB/s vs HBM peak",
        "value": round(B * args.steps / elapsed, 1), "unit": "seed nodes/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}-shaped synthetic: N={N} E={E} F={Fdim} C={C}, B={B}/GPU, fan-out {K}, "
                               f"hidden {args.hidden}, aggregator {args.aggregator}, Adam lr {args.lr}, TRAINER step "
                               "(supervised + unsupervised term, 3 forwards, module path)",
                   "parallelism": "dp1", "global_batch": B},
        "roofline": None, "cpu_baseline": None, "final_loss": round(float(loss.detach()), 5), "variant": "trainer-module"}), flush=True)


MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # MI355X_MICROARCH.md: dense peaks


def bracket_contractions(ops):
    """Wrap the contraction entry points of graphsage_amd.ops: while rec["on"], the MFMA kernel of
    every call above 1 GFLOP carries a HIP event pair (sage_probe_next_contraction: start / stop
    events attached to the launch — the dispatch's own duration; rec["bracket"] = True records the
    pair around the call instead, which adds the markers' own cost).  Returns rec (flops, pairs)."""
    import ctypes

    from graphsage_amd import _lib
    rec = {"on": False, "flops": 0.0, "pairs": [], "bracket": False}

    def bracket(name, flops_of):
        inner = getattr(ops, name)

        def wrapped(*a, **kw):
            if not rec["on"]:
                return inner(*a, **kw)
            fl = flops_of(*a, **kw)
            if fl < 1e9:
                return inner(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if rec["bracket"]:
                out = inner(*a, **kw)
                e1.record()
            else:
                e1.record()          # (torch creates the hipEvent_t on the first record)
                lib = _lib.load()
                _lib.check(lib.sage_probe_next_contraction(ctypes.c_void_p(e0.cuda_event), ctypes.c_void_p(e1.cuda_event), 0),
                           "sage_probe_next_contraction")
                try:
                    out = inner(*a, **kw)
                finally:
                    lib.sage_probe_next_contraction(None, None, 0)   # (a pair no launch took is dropped)
            rec["flops"] += fl
            rec["pairs"].append((e0, e1))
            return out
        setattr(ops, name, wrapped)

    def rows_of(m, m_dev, default):
        mm = m if m is not None else default
        # a device-side count bounds the rows that are really contracted (reading it synchronises:
        # measurement runs only)
        return min(mm, int(m_dev)) if m_dev is not None else mm

    def fwd_flops(a1, w1, a2=None, w2=None, m=None, m_dev=None, rows1=None, **kw):
        mm = rows_of(m, m_dev, rows1.numel() if rows1 is not None else a1.size(0))
        return 2.0 * mm * w1.size(0) * (w1.size(1) + (w2.size(1) if w2 is not None else 0))
    bracket("linear_concat_fwd", fwd_flops)
    bracket("pool_linear_fwd", lambda x, w, bias, m=None, m_dev=None, rows=None, **kw:
            2.0 * rows_of(m, m_dev, rows.numel() if rows is not None else x.size(0)) * w.size(0) * w.size(1))
    bracket("linear_dweight", lambda dout, a, m=None, m_dev=None, **kw:
            2.0 * rows_of(m, m_dev, dout.size(0)) * dout.size(1) * a.size(1))
    bracket("linear_dweight2", lambda dout, a1, dw1, a2, dw2, m=None, m_dev=None, **kw:
            2.0 * rows_of(m, m_dev, dout.size(0)) * dout.size(1) * (a1.size(1) + a2.size(1)))
    bracket("linear_dinput", lambda dout, w, m=None, m_dev=None, **kw:
            2.0 * rows_of(m, m_dev, dout.size(0)) * w.size(0) * w.size(1))
    bracket("linear_dinput2", lambda dout, w1, out1, w2, out2, m=None, m_dev=None, **kw:
            2.0 * rows_of(m, m_dev, dout.size(0)) * w1.size(0) * (w1.size(1) + w2.size(1)))
    return rec


def mfma_roofline(rec, steps, gemm_mode):
    t_ms = sum(a.elapsed_time(b) for a, b in rec["pairs"])
    tf = rec["flops"] / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0
    # what an event pair with nothing between its records reads on this stream (the bracket's own cost)
    eb = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a0, b0 in eb:
        a0.record()
        b0.record()
    torch.cuda.synchronize()
    empty_ms = sum(a0.elapsed_time(b0) for a0, b0 in eb) / len(eb)
    # bf16x3: six bf16 MFMA products per fp32 product, so the fp32-equivalent peak is bf16 / 6
    peak = MFMA_PEAK_TFLOPS["bf16"] / 6.0 if gemm_mode == "bf16x3" else MFMA_PEAK_TFLOPS["f32"]
    return {"bound": "mfma", "kernel": "MFMA contractions above 1 GFLOP (gemm_x3_kernel / gemm_x3_dw_kernel / gemm_kernel)",
            "achieved": round(tf, 1), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(tf / peak, 4),
            "traffic": None, "flops_per_step": rec["flops"] / steps,
            "gemm_ms_per_step": round(t_ms / steps, 4), "launches_per_step": len(rec["pairs"]) / steps,
            "avg_launch_us": round(t_ms / max(len(rec["pairs"]), 1) * 1e3, 2),
            "empty_event_pair_us": round(empty_ms * 1e3, 2),
            "timing": ("HIP events recorded around each call on its stream" if rec.get("bracket") else
                       "HIP start/stop events attached to each contraction's launch (hipExtLaunchKernelGGL)")
                      + " (fp32-equivalent flops; peak = bf16 dense peak / 6 products in bf16x3 mode)"}


def module_variant(args, data, graph, model, N, E, Fdim, C, n_train, B, K, gemm_mode):
    """Aggregators the static-shape engine does not run (lstm; the pooling ones with
    --module-path): fig2a's step through the module path (autograd Functions over the same
    kernels).  `roofline` is for the dominant kernels here, the large MFMA contractions: every
    sage_linear_concat_fwd / sage_linear_dweight(2) call above 1 GFLOP is bracketed by a HIP
    event pair on its stream; achieved = sum of their flops / sum of their event times."""
    import torch.nn.functional as F

    from graphsage_amd import ops
    from graphsage_amd.samplers import UniformLoader
    gen = torch.Generator().manual_seed(0)
    train = torch.randperm(N, generator=gen)[:n_train]
    loader = UniformLoader(data, num_neighbors=K, input_nodes=train, batch_size=B, shuffle=False, seed=0)
    # (one fused kernel over all parameters instead of torch's default six foreach passes: the lstm
    # weights are 145 MB, the passes 0.7 of 11.4 ms per step)
    opt = torch.optim.Adam(model.parameters(), lr=args.lr, fused=not os.environ.get("SAGE_BENCH_FOREACH_ADAM"))
    rec = bracket_contractions(ops)

    def step(batch):
        opt.zero_grad()
        y_hat = model.forward_batch(batch)
        loss = F.cross_entropy(y_hat, batch.y[:batch.batch_size])
        loss.backward()
        opt.step()
        return loss

    it = iter(loader)
    for _ in range(args.warmup):
        step(next(it))
    torch.cuda.synchronize()
    rec["on"] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(next(it))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    extra = f", LSTM_NUM_INPUTS {args.lstm_num_inputs}" if args.aggregator == "lstm" else ""
    print(json.dumps({
        "metric": "Reddit GraphSAGE-mean 25/10 fwd+bwd seed nodes/s; gather GB/s vs HBM peak",
        "value": round(B * args.steps / elapsed, 1), "unit": "seed nodes/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}-shaped synthetic: N={N} E={E} F={Fdim} C={C}, B={B}/GPU, fan-out {K}, "
                               f"hidden {args.hidden}, aggregator {args.aggregator}{extra}, Adam lr {args.lr}, fig2a step "
                               "(supervised term only), module path",
                   "parallelism": "dp1", "global_batch": B},
        "roofline": mfma_roofline(rec, args.steps, gemm_mode),
        "cpu_baseline": None, "gemm_mode": gemm_mode, "final_loss": round(float(loss.detach()), 5),
        "variant": "module-path"}), flush=True)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the
    N ranks as fresh child processes through torch.distributed.run and relay their output and
    exit code.  This process has not touched the GPU (importing torch does not initialise it) and
    never replaces itself — the ranks are children, rank 0's JSON line goes to the inherited
    stdout."""
    import socket
    import subprocess
    with socket.socket() as s:          # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this host driver
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (graphsage_amd has no CPU path)")
    if os.environ.get("SAGE_BENCH_WATCHDOG"):
        # debugging aid: dump every thread's Python stack and exit if the run takes this long
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SAGE_BENCH_WATCHDOG"]), exit=True)
    if world != args.gpus:
        # n_gpus in the result is the number of ranks that really ran: refuse a launch whose
        # world size is not what --gpus says (e.g. `python bench.py --gpus 8` without torchrun)
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: launch it with "
                         f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...`")
    # Rehearsal of the N > 1 path on a ONE-GPU box (never what the driver runs): with
    # SAGE_BENCH_BACKEND=gloo every rank uses cuda:0 and the gradient all-reduce goes over gloo
    # (RCCL refuses two ranks on one device).  The number it prints is not a scaling result.
    backend = os.environ.get("SAGE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"{world} ranks on a node with {torch.cuda.device_count()} visible GPUs: RCCL needs one GPU per rank")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_coll = args.force_collective and world == 1
    # RCCL prints a version banner on the process's stdout when its first communicator comes up:
    # with a process group, everything that is not the result goes to stderr (fd level, so that
    # native libraries are covered too) and rank 0 writes its ONE JSON line to the real stdout
    result_fd = None
    if world > 1 or force_coll:
        sys.stdout.flush()
        result_fd = os.dup(1)
        os.dup2(2, 1)
    if force_coll:
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
        if torch.distributed.get_world_size() != world:
            raise SystemExit("process group came up with fewer ranks than WORLD_SIZE")

    from graphsage_amd import ops, settings
    from graphsage_amd.datasets import reddit_like_device
    from graphsage_amd.engine import _REDUCE, SageTrainEngine
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import NeighborSampler
    from graphsage_amd.samplers.uniform_loader import device_graph

    N, E, Fdim, C, n_train = SHAPES[args.workload]
    B, K = args.batch_size, list(args.fanout)
    t_setup = time.perf_counter()
    gemm_mode = "f32" if os.environ.get("SAGE_GEMM_X3", "")[:1] == "0" else "bf16x3"
    data = reddit_like_device(N, E, Fdim, C, seed=0, device=dev)   # identical on every rank
    multilabel = args.workload == "ppi"
    if multilabel:
        gl = torch.Generator(device=dev).manual_seed(1)
        data.y = (torch.rand((N, C), generator=gl, device=dev) < 0.3).float()
    graph = device_graph(data)
    feats, labels = graph.features(data), graph.labels(data)
    torch.manual_seed(0)
    if args.variant == "unsup":
        C = args.hidden              # the unsupervised runners emit embeddings, not class logits
    if args.aggregator == "lstm":
        settings.LSTM_NUM_INPUTS = args.lstm_num_inputs
    model = GraphSAGE(Fdim, args.hidden, args.aggregator, len(K), C).to(dev)
    # (the aggregators whose dominant kernels are contractions: MFMA roofline instead of the gather's)
    pooling = args.aggregator in ("max_pool", "mean_pool", "lstm")
    on_engine = (args.aggregator in _REDUCE and args.variant != "trainer-module"
                 and not (pooling and (args.module_path or args.variant != "fig2a")))
    if not on_engine and args.variant == "fig2a":
        if world != 1:
            raise SystemExit("module-path aggregators are single-GPU measurements here")
        return module_variant(args, data, graph, model, N, E, Fdim, C, n_train, B, K, gemm_mode)
    data.edge_index = None                                          # CSC is resident; free the COO
    torch.cuda.empty_cache()
    init_layers = [{k: getattr(m, k).weight.detach().cpu().clone() for k in ("lin_l", "lin_r") if hasattr(m, k)}
                   for m in model.layers]
    pg = torch.distributed.group.WORLD if (world > 1 or force_coll) else None
    if args.variant == "trainer-module":
        return trainer_variant(args, data, graph, model, N, E, Fdim, C, n_train, B, K, world, rank)
    unsup = {"trainer": True, "unsup": "only"}.get(args.variant, False)

    # seeds: first n_train ids of randperm(N, seed 0); rank r takes the strided shard r::world
    gen = torch.Generator().manual_seed(0)
    train = torch.randperm(N, generator=gen)[:n_train]
    mine = train[rank::world].to(dev)
    n_batches = mine.numel() // B
    if n_batches == 0:
        raise SystemExit("workload too small for this batch size / world size")

    def seeds_of(step):
        b = step % n_batches
        return mine[b * B:(b + 1) * B]

    def counter_of(step):
        return step * world + rank

    use_graph = args.graph and not args.no_graph
    # the hbm roofline probe times the layer-0 gather; with a pooling aggregator the dominant
    # kernels are the contractions (MFMA roofline below) and the gather reads pool outputs
    probe = not args.no_probe and not pooling
    attached = probe and not args.bracket_probe

    def make_engine(mdl, overlap):
        return SageTrainEngine(mdl, NeighborSampler.from_graph(graph, K, seed=0), feats, labels, B, lr=args.lr,
                               world_size=world, process_group=pg, overlap_gather=overlap, group=args.group,
                               unsup=unsup, force_collective=force_coll)

    def run_schedule(engine, settle, warmup, steps, probe_gather, graph_replay=False):
        """settle + warmup untimed steps, then `steps` timed ones bracketed by barrier + synchronize;
        with probe_gather one HIP event pair per timed step is attached to the launch of the layer-0
        gather (hipExtLaunchKernelGGL start / stop events: that kernel's own duration on its
        stream; --bracket-probe records the pair around the launch instead)."""
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)] \
            if probe_gather else []
        handles = []
        if probe_gather and attached:
            for a0, b0 in ev:          # torch creates the hipEvent_t on the first record
                a0.record()
                b0.record()
            handles = [(a0.cuda_event, b0.cuda_event) for a0, b0 in ev]
        # the two large contractions of the fused mean / sum step carry event pairs of their own in
        # the same run (sage_probe_next_contraction: per step the forward is the first MFMA
        # contraction this thread launches, the weight-gradient pair the second)
        probe_mfma = (probe_gather and attached and not graph_replay and not engine.overlap_gather
                      and engine.fused_logit and engine.L == 2)
        ev_mm, h_mm = [], []
        if probe_mfma:
            import ctypes

            from graphsage_amd import _lib
            plib = _lib.load()
            ev_mm = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(steps)]
            for quad in ev_mm:
                for e in quad:
                    e.record()
            h_mm = [tuple(ctypes.c_void_p(e.cuda_event) for e in quad) for quad in ev_mm]
        cur = {"i": 0}
        if graph_replay:
            engine.capture(mine, counter0=rank, counter_stride=world, split_gather=probe_gather)
            before = (lambda: ev[cur["i"]][0].record()) if (probe_gather and not attached) else None
            after = (lambda: ev[cur["i"]][1].record()) if (probe_gather and not attached) else None

            def run_step(s, timed):
                if timed and probe_gather:
                    engine.replay(before, after, handles[cur["i"]] if attached else None)
                else:
                    engine.replay()
        else:
            depth = engine.depth
            for d in range(depth):
                engine.prefetch(seeds_of(d), counter_of(d))

            def run_step(s, timed):
                pg_ = probe_gather and timed
                b = (lambda: ev[cur["i"]][0].record()) if (pg_ and not attached) else None
                a = (lambda: ev[cur["i"]][1].record()) if (pg_ and not attached) else None
                if probe_mfma and pg_:
                    f0, f1, d0, d1 = h_mm[cur["i"]]
                    plib.sage_probe_next_contraction(f0, f1, 0)
                    plib.sage_probe_next_contraction(d0, d1, 1)
                engine.pipelined_step(seeds_of(s + depth), counter_of(s + depth), b, a,
                                      handles[cur["i"]] if (attached and pg_) else None)
        for s in range(settle):
            cur["i"] = s % max(steps, 1)
            run_step(s, s < 4 * engine.nslots)        # includes the probe's launch path: its tapes exist
        for s in range(settle, settle + warmup):
            run_step(s, False)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        st0 = engine.state_sum.clone()
        t0 = time.perf_counter()
        for s in range(settle + warmup, settle + warmup + steps):
            cur["i"] = s - warmup - settle
            run_step(s, True)
        engine.flush()                                 # data parallel: the last step's deferred Adam
        t_issued = time.perf_counter() - t0            # host-side enqueue time of the timed steps
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(t)
        if not graph_replay:       # drain the pipeline so that another engine can follow
            while engine.primed:
                engine.pipelined_step()
            engine.flush()
            torch.cuda.synchronize()
        st = (engine.state_sum - st0).cpu().double()       # sums over the timed steps (+ the drain's, unused)
        if int(engine.state_sum[ops.ST_ERROR]) != 0:
            raise SystemExit("sampler reported an error during the run")
        ms = torch.tensor([a.elapsed_time(b) for a, b in ev], dtype=torch.float64) if probe_gather else None
        mm = None
        if probe_mfma:
            plib.sage_probe_next_contraction(None, None, 0)
            mm = (sum(q[0].elapsed_time(q[1]) for q in ev_mm), sum(q[2].elapsed_time(q[3]) for q in ev_mm))
        return {"elapsed": elapsed, "issued": t_issued, "st": st, "gather_ms": ms, "loss": float(engine.loss),
                "mfma_ms": mm}

    def gather_bytes(st, launches):
        e_b = float(st[ops.ST_EDGES])                    # summed over the timed launches
        t1 = float(st[ops.ST_HOP_NODES + 1])
        return 4.0 * Fdim * e_b + 4.0 * Fdim * t1 + 8.0 * (2 * e_b + t1 + launches)

    base = max(args.settle_steps, 0)
    # One GPU: the overlapped schedule.  Data parallel: the serial one — the layer-0 gather of
    # batch s+1 then sits on the main stream between the all-reduce launch of step s and its
    # Adam step and covers the collective (graphsage_amd/engine.py).
    # (narrow rows — PPI's 50 floats — make every kernel latency-bound: the overlapped schedule's
    # cross-stream hand-shake then costs more than the gather it hides, 0.187 against 0.158 ms/step;
    # the three-forward steps run faster serially too: the engine's overlap_gather='auto' rule,
    # which the trainers use as well)
    overlapped = (not args.serial and not use_graph and not pooling and world == 1 and not force_coll
                  and (args.overlapped or SageTrainEngine.gather_pays_overlap(Fdim, B, K, bool(unsup))))
    import copy
    # 1. the roofline probe: serial schedule (every kernel of a step on ONE stream), where the
    #    HBM-bound layer-0 gather has the chip to itself and its event-timed duration is its own
    #    (also when the headline schedule is the serial one — narrow rows, data parallel: the event
    #    pairs then stay out of the timed region that `value` comes from; on PPI-sized steps they cost
    #    10 us per step, and at N > 1 they would count against the scaling)
    probe_res = None
    if probe and not use_graph:
        pm = copy.deepcopy(model)
        probe_res = run_schedule(make_engine(pm, False), base, args.warmup, max(1, min(args.probe_steps, args.steps)), True)
        base_main = min(base, 300)       # the process is warm now
    else:
        base_main = base
    # 2. the headline run
    engine = make_engine(model, overlapped)
    setup_s = time.perf_counter() - t_setup
    #    (no event pairs in its timed region: the products-shaped step loses 0.02 ms to them, the
    #    headline 0.002; only the graph replay, which has no separate probe run, carries them)
    res = run_schedule(engine, base_main, args.warmup, args.steps, probe and use_graph, graph_replay=use_graph)
    elapsed, st, loss = res["elapsed"], res["st"], res["loss"]
    # the drain after the timed region sampled nothing new, but keep the counts exact anyway:
    # ST sums cover the timed steps because prefetches of the drain are not issued (no next_seeds)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        roofline = roofline_mfma = roofline_mfma_serial = None
        if world == 1 and not args.no_probe and not force_coll and args.variant == "fig2a":
            # MFMA roofline of the contractions: a short run of the same steps with every call
            # going through the wrappers of ops.py (no call tapes), each contraction above
            # 1 GFLOP bracketed by a HIP event pair
            rec = bracket_contractions(ops)
            pe = make_engine(copy.deepcopy(model), False)
            pe.use_tapes = False
            pe.pool_fork = None       # every contraction alone on the chip: its own duration
            n_probe = max(1, min(20, args.steps))
            for s_ in range(3):
                pe.step(seeds_of(s_), counter_of(s_))
            torch.cuda.synchronize()
            rec["on"] = True
            for s_ in range(3, 3 + n_probe):
                pe.step(seeds_of(s_), counter_of(s_))
            torch.cuda.synchronize()
            rec["on"] = False
            roofline_mfma = mfma_roofline(rec, n_probe, gemm_mode) if rec["pairs"] else None   # (none above 1 GFLOP: PPI)
            if roofline_mfma is not None:
                roofline_mfma["schedule"] = (f"separate run of {n_probe} steps on one stream without call tapes "
                                             "(every contraction alone on the chip)")
            del pe
            if pooling:      # there the contractions ARE the dominant kernels
                roofline, roofline_mfma = roofline_mfma, None
        if probe:
            src = probe_res if probe_res is not None else res
            n_launch = src["gather_ms"].numel()
            # the timed steps' sampled batches: in the pipelined loop the sampler runs `depth` batches
            # ahead, so the state sums and the gathers timed cover batch sets shifted by `depth` of
            # the same stationary stream; both are averages over >= 300 batches
            bytes_sum = gather_bytes(src["st"], n_launch) * (n_launch / max(float(src["st"][ops.ST_HOP_NODES]) / B, 1.0))
            ms = src["gather_ms"]
            achieved = float(bytes_sum / (float(ms.sum()) * 1e-3) / 1e9)
            traffic, traffic_src = None, None
            tpath = os.path.join(ROOT, "profiles", "gather_traffic.json")
            if args.workload == "reddit" and os.path.exists(tpath):
                # HBM bytes per launch from rocprofv3 PMC passes of this same kernel on this workload
                # (FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 correction applied) — measured
                # offline, see the file
                with open(tpath) as f:
                    tj = json.load(f)
                traffic, traffic_src = tj["traffic_bytes_per_launch"], "profiles/gather_traffic.json"
            eb = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            for a0, b0 in eb:
                a0.record()
                b0.record()
            torch.cuda.synchronize()
            bracket_us = sum(a0.elapsed_time(b0) for a0, b0 in eb) / len(eb) * 1e3
            roofline = {"bound": "hbm", "kernel": f"gather_reduce_fwd_kernel (layer 0, {args.aggregator}, node_map)",
                        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "avg_launch_us": round(float(ms.mean()) * 1e3, 2),
                        "schedule": ("serial probe run of this same workload (%d timed steps, every kernel on one stream: "
                                     "the gather alone on the chip)" % n_launch) if probe_res is not None
                                    else ("serial headline run" if not use_graph else "graph replay, gather launched eagerly between two graphs"),
                        "timing": ("HIP start/stop events attached to each launch (hipExtLaunchKernelGGL)" if attached
                                   else "HIP events recorded around each launch"),
                        "empty_event_pair_us": round(bracket_us, 2),
                        "algorithmic_bytes_per_launch": int(bytes_sum / n_launch), "row_stride_floats": int(feats.stride(0))}
            if src.get("mfma_ms") is not None:
                # MFMA roofline of the step's two large contractions from the SAME serial run (taped,
                # steady clocks): flops from the device-side row counts summed over those launches
                t1_sum = float(src["st"][ops.ST_HOP_NODES + 1]) * (n_launch / max(float(src["st"][ops.ST_HOP_NODES]) / B, 1.0))
                fl_each = 2.0 * t1_sum * args.hidden * 2 * Fdim
                f_ms, d_ms = src["mfma_ms"]
                peak = MFMA_PEAK_TFLOPS["bf16"] / 6.0 if gemm_mode == "bf16x3" else MFMA_PEAK_TFLOPS["f32"]
                tf = 2.0 * fl_each / ((f_ms + d_ms) * 1e-3) / 1e12
                roofline_mfma_serial = {
                    "bound": "mfma", "kernel": "layer-0 forward contraction + layer-0 weight-gradient pair "
                                               "(gemm_x3_kernel<64> / gemm_x3_dw_kernel<128>; gemm_kernel with SAGE_GEMM_X3=0)",
                    "achieved": round(tf, 1), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                    "traffic": None, "flops_per_step": 2.0 * fl_each / n_launch,
                    "forward_avg_launch_us": round(f_ms / n_launch * 1e3, 2),
                    "dweight_avg_launch_us": round(d_ms / n_launch * 1e3, 2),
                    "forward_tflops": round(fl_each / (f_ms * 1e-3) / 1e12, 1),
                    "dweight_tflops": round(fl_each / (d_ms * 1e-3) / 1e12, 1),
                    "schedule": roofline["schedule"],
                    "timing": "HIP start/stop events attached to each contraction's launch (hipExtLaunchKernelGGL); "
                              "fp32-equivalent flops; peak = bf16 dense peak / 6 products in bf16x3 mode"}
            if probe_res is not None:
                # the same kernel in the headline (overlapped) schedule, where it shares the chip with
                # the MFMA-bound forward GEMM of the previous batch: longer by contention, not by itself
                roofline["serial_ms_per_step"] = round(probe_res["elapsed"] / n_launch * 1e3, 4)
                if overlapped:
                    # (a short run of the headline schedule with the event pairs attached, behind the timed one)
                    n_ov = max(1, min(200, args.steps))
                    r_ov = run_schedule(make_engine(copy.deepcopy(model), True), 50, args.warmup, n_ov, True)
                    roofline["overlapped_avg_launch_us"] = round(float(r_ov["gather_ms"].mean()) * 1e3, 2)
        # Informational only (never `value`): the headline schedule with the big contractions on the
        # exact-fp32 MFMA (SAGE_GEMM_X3=0) instead of the default bf16x3 split
        variant_f32 = None
        if world == 1 and not use_graph and not args.no_variant and args.variant == "fig2a" and gemm_mode == "bf16x3":
            os.environ["SAGE_GEMM_X3"] = "0"      # read by the library at every call
            try:
                m2 = copy.deepcopy(model)
                nv = min(args.steps, 200)
                r2 = run_schedule(make_engine(m2, overlapped), 100, args.warmup, nv, False)
            finally:
                del os.environ["SAGE_GEMM_X3"]
            variant_f32 = {"gemm": "exact-fp32 MFMA contractions (SAGE_GEMM_X3=0)", "steps": nv,
                           "ms_per_step": round(r2["elapsed"] / nv * 1e3, 4),
                           "value": round(B * nv / r2["elapsed"], 1), "unit": "seed nodes/s"}
        cpu = None
        if world == 1 and args.cpu_steps > 0 and args.variant == "fig2a" and args.aggregator == "mean" and not multilabel:   # the oracle times fig2a's step
            x_cpu, y_cpu = data.x.cpu(), data.y.cpu()
            colptr, row = graph.colptr.cpu().numpy(), graph.row.cpu().numpy()
            seeds_list = [train[i * B:(i + 1) * B].numpy() for i in range(args.cpu_steps + args.cpu_warmup)]
            v, cores, per_step, n_timed, mean_step = cpu_baseline((x_cpu, y_cpu), colptr, row, seeds_list, K,
                                                                  init_layers, args.cpu_steps, args.lr, args.cpu_warmup)
            cpu = {"value": round(v, 1), "unit": "seed nodes/s", "cores": cores, "kind": "port",
                   "cpu_model": cpu_model(),
                   "sample": f"{n_timed} timed steps after {args.cpu_warmup} warm-up steps of the same workload, B={B}, "
                             f"median {per_step:.3f} s/step (mean {mean_step:.3f}), oracle restatement of the "
                             "reference step as written"}
        out = {
            "metric": "Reddit GraphSAGE-mean 25/10 fwd+bwd seed nodes/s; gather GB/s vs HBM peak",
            "value": round(value, 1), "unit": "seed nodes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}-shaped synthetic: N={N} E={E} F={Fdim} C={C}, "
                                   f"B={B}/GPU, fan-out {K}, hidden {args.hidden}, aggregator {args.aggregator}"
                                   + (f" (LSTM_NUM_INPUTS {args.lstm_num_inputs})" if args.aggregator == "lstm" else "") + ", "
                                   f"Adam lr {args.lr}, " + ("multi-label BCE-with-logits, " if multilabel else "")
                                   + {"fig2a": "fig2a step (supervised term only)",
                                                       "unsup": "UNSUPERVISED trainer step (random-walk loss, 3 forwards)"}.get(
                                       args.variant, "TRAINER step (supervised + unsupervised term, 3 forwards)"),
                       "parallelism": f"dp{world}", "global_batch": world * B},
            "roofline": roofline,
            # the contractions: from the serial probe run's attached events where that run has them
            # (the fused mean / sum step), else from the separate untaped run
            "roofline_mfma": roofline_mfma_serial if roofline_mfma_serial is not None else roofline_mfma,
            "roofline_mfma_untaped": roofline_mfma if roofline_mfma_serial is not None else None,
            "cpu_baseline": cpu,
            "schedule": ("HIP-graph replay" if use_graph else
                         "overlapped: layer-0 gather of batch s+1 on the side stream beside batch s" if overlapped
                         else "serial: every kernel of a step on one stream" if (world == 1 and not force_coll)
                         else "data parallel, serial: all-reduce launch (RCCL stream) -> layer-0 gather of the next "
                              "batch -> wait -> Adam -> forward contraction"),
            "gemm_mode": gemm_mode + (" (fp32 operands split into 3 bf16 terms, 6 MFMA products, fp32 accumulation)"
                                      if gemm_mode == "bf16x3" else " (v_mfma_f32_32x32x2_f32)"),
            "variant_gemm_f32": variant_f32,
            "final_loss": round(loss, 5), "setup_s": round(setup_s, 1), "settle_steps": base_main,
            "avg_nodes_per_batch": float(st[ops.ST_NODES]) / max(float(st[ops.ST_HOP_NODES]) / B, 1.0),
            "avg_edges_per_batch": float(st[ops.ST_EDGES]) / max(float(st[ops.ST_HOP_NODES]) / B, 1.0),
            "variant": args.variant, "hip_graph": use_graph, "force_collective": bool(force_coll),
            "host_enqueue_ms_per_step": round(res["issued"] / args.steps * 1e3, 4), "overlap_gather": bool(overlapped),
        }
        if result_fd is None:
            print(json.dumps(out), flush=True)
        else:
            os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or force_coll:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
