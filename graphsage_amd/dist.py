"""Data-parallel helpers (one process per GPU, torch.distributed; backend nccl = RCCL over
xGMI on the GPU box, gloo in the CPU tests).  The reference has no multi-GPU code at all
(graphsage/settings.py:32 selects a single DEVICE); this is the one strategy SURVEY §8e adds:
replicated graph + features, seed nodes sharded across ranks, ONE all-reduce of a flat
gradient buffer per step."""
import torch
import torch.distributed as dist


def shard_seeds(index, rank, world_size):
    """Strided shard with equal counts on every rank (the tail that does not divide is dropped)."""
    if world_size <= 1:
        return index
    per = index.numel() // world_size
    return index[: per * world_size][rank::world_size]


def batch_counter(epoch, batches_per_epoch, batch, rank, world_size):
    """Philox batch counter: distinct for every (epoch, batch, rank)."""
    return (epoch * batches_per_epoch + batch) * world_size + rank


class _Done:
    """Work-like handle of a collective that has already completed on the caller's stream."""

    def wait(self):
        return True


_staging = {}


def _all_reduce_staged(flat_grad, group):
    """gloo with a DEVICE tensor (tests and rehearsals with several ranks on one GPU; the
    production backend is nccl): stage through pinned host memory on the caller's stream and
    run the collective on the host copy.  ProcessGroupGloo's own device path hung here with two
    ranks sharing a GPU (round 1; gpurun_out/dp_sync.log: between steps 30 and 40 one rank sits
    in torch.cuda.synchronize(), the other in work.wait() on the collective the first has not
    joined; with GPU_MAX_HW_QUEUES=32, dp_sync_q32.log, every ten steps take twice as long as
    the ten before and it ends the same way).  Cause undetermined — DESIGN.md section 4."""
    key = (flat_grad.numel(), flat_grad.dtype)
    host = _staging.get(key)
    if host is None:
        host = _staging[key] = torch.empty(flat_grad.numel(), dtype=flat_grad.dtype, pin_memory=True)
    host.copy_(flat_grad.reshape(-1), non_blocking=True)
    torch.cuda.current_stream().synchronize()
    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
    flat_grad.reshape(-1).copy_(host, non_blocking=True)     # ordered on the current stream


def all_reduce_flat(flat_grad, group=None, async_op=False, force=False):
    """Sum the flat gradient buffer over ranks (the mean's 1/world is fused into Adam).
    async_op=True returns the work handle (None when there is nothing to reduce): the collective
    runs on the backend's own stream and handle.wait() orders the current stream after it.
    force: issue the collective in a one-rank group as well (the engine's force_collective)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        if flat_grad.is_cuda and dist.get_backend(group) == "gloo":
            _all_reduce_staged(flat_grad, group)
            return _Done() if async_op else flat_grad
        work = dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return work if async_op else flat_grad
    return None if async_op else flat_grad


class PendingReduce:
    """The data-parallel step's deferred half: ONE gradient all-reduce in flight whose optimizer
    step is applied later (graphsage_amd/engine.py: launched after the backward of step i,
    completed just before the first kernel of step i+1 that reads the weights, so the collective
    overlaps the sampler hand-shake and the weight-free layer-0 gather in between).  Works on
    whatever all_reduce_flat works on; tests/test_dist_cpu.py drives it with gloo on CPU tensors
    (a real asynchronous work handle kept across a step)."""

    def __init__(self):
        self.work = None
        self.active = False
        self._done = None       # device path: event recorded on the collective's stream

    def launch(self, flat_grad, group=None, force=False, stream=None):
        """stream (device tensors): a stream of the CALLER's choosing that carries the collective.
        The collective is issued there as a synchronous op — which ProcessGroupNCCL enqueues on
        the current stream, with no stream or event of its own — between a wait for the launching
        stream and an event that complete() makes its stream wait for.  The caller thereby picks
        the hardware queue (graphsage_amd/engine.py probes for one that runs beside its compute
        streams: a stream from the backend's pool can share the main stream's queue, which puts
        the collective back on the critical path), and the host pays one cheap call instead of an
        asynchronous work handle (7.6 against 25.6 us on one rank, scratch/nccl_host_cost.py)."""
        if self.active:
            raise RuntimeError("the previous step's all-reduce has not been completed")
        if stream is not None and flat_grad.is_cuda:
            import torch
            cur = torch.cuda.current_stream(flat_grad.device)
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                all_reduce_flat(flat_grad, group, async_op=False, force=force)
            if self._done is None:
                self._done = torch.cuda.Event()
            self._done.record(stream)
            self.work = self._done
        else:
            self.work = all_reduce_flat(flat_grad, group, async_op=True, force=force)
        self.active = True

    def complete(self):
        """True if a reduce was pending; on return the caller (the current stream, for device
        tensors) is ordered after the collective and may read the summed gradient."""
        if not self.active:
            return False
        work, self.work, self.active = self.work, None, False
        if work is self._done and work is not None:
            import torch
            torch.cuda.current_stream().wait_event(work)
        elif work is not None:
            work.wait()
        return True


def all_reduce_gradients(model, group=None):
    """Module path under data parallelism (any aggregator, any loss): average the gradients of
    `model` over ranks with ONE all-reduce of a flat buffer, in place — call it between
    loss.backward() and optimizer.step().  Parameters without a gradient on this rank count as
    zeros, so every rank reduces the same layout.  No-op outside a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        return
    params = [p for p in model.parameters() if p.requires_grad]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    all_reduce_flat(flat, group)
    flat.div_(dist.get_world_size(group))
    off = 0
    for p in params:
        n = p.numel()
        if p.grad is None:
            p.grad = flat[off:off + n].view_as(p).clone()
        else:
            p.grad.copy_(flat[off:off + n].view_as(p))
        off += n
