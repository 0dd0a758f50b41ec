"""Tensor-level wrappers over the C ABI (include/sage_hip.h).

Each function validates shapes/dtypes/devices on the host (a wrong shape handed to a
hand-written kernel is a GPU fault, not an exception), takes raw device pointers from the
torch tensors and enqueues the kernel on torch's current stream.  torch is plumbing here:
it owns the memory and the stream, the arithmetic is in libsage_hip.so.
"""
import ctypes
from ctypes import c_void_p

import torch

from . import _lib

REDUCE = {"sum": 0, "add": 0, "mean": 1, "max": 2, "gcn": 3}
EPI_NONE, EPI_BIAS, EPI_RELU, EPI_ACCUM, EPI_ZEROED = 0, 1, 2, 4, 8

ST_NODES, ST_EDGES, ST_FRONT_BEGIN, ST_FRONT_END, ST_ERROR = 0, 1, 2, 3, 4
ST_HOP_NODES, ST_HOP_EDGES, STATE_WORDS = 8, 24, 40
CTRL_COUNTER, CTRL_STAMP, CTRL_SEED_OFFSET, CTRL_WORDS = 0, 1, 2, 4


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, dtype, name, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise RuntimeError(f"{name}: tensor required")
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA (HIP) tensor; graphsage_amd has no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return c_void_p(t.data_ptr())


def new_event(stream=None):
    """A torch event whose HIP handle exists (torch creates it at the first record)."""
    ev = torch.cuda.Event()
    ev.record(stream if stream is not None else torch.cuda.current_stream())
    return ev


def event_record(ev, stream):
    """hipEventRecord(ev, stream) as a library call: an entry of a recorded call tape."""
    rc = _lib.load().sage_event_record(c_void_p(ev.cuda_event), c_void_p(stream.cuda_stream))
    _lib.check(rc, "sage_event_record")


def stream_wait_event(stream, ev):
    """hipStreamWaitEvent(stream, ev) as a library call: an entry of a recorded call tape."""
    rc = _lib.load().sage_stream_wait_event(c_void_p(stream.cuda_stream), c_void_p(ev.cuda_event))
    _lib.check(rc, "sage_stream_wait_event")


def empty_rows(m, n, device, dtype=torch.float32, zero=False):
    """[m, n] matrix whose rows start 64-byte aligned: a view of a buffer with the row stride
    rounded up to 16 elements when n is not a multiple of 4 (e.g. 602 -> 608), so that every
    kernel takes its 16-byte vector path on it (the 8-byte variants of the GEMM are ~1.7x slower)."""
    make = torch.zeros if zero else torch.empty
    if n % 4 == 0 or dtype != torch.float32:
        return make((m, n), dtype=dtype, device=device)
    ld = (n + 15) // 16 * 16
    return make((m, ld), dtype=dtype, device=device)[:, :n]


def aligned_rows(t):
    """t itself if its rows can be read 16 bytes at a time, else a row-padded copy (empty_rows)."""
    if t.dim() != 2 or t.dtype != torch.float32:
        return t
    if t.stride(1) == 1 and t.data_ptr() % 16 == 0 and (t.size(0) <= 1 or t.stride(0) % 4 == 0):
        return t
    out = empty_rows(t.size(0), t.size(1), t.device)
    out.copy_(t)
    return out


def _mat(t, name):
    """2-D fp32 matrix with unit column stride -> (pointer, row stride)."""
    if t.dim() != 2 or (t.size(1) > 1 and t.stride(1) != 1):
        raise RuntimeError(f"{name}: expected a 2-D tensor with contiguous rows, got "
                           f"shape {tuple(t.shape)} strides {t.stride()}")
    ld = t.stride(0) if t.size(0) > 1 else max(t.stride(0), t.size(1))
    return _p(t, torch.float32, name), int(ld)


_dummies = {}


def _vec(t, dtype, name, min_len=0, allow_none=False):
    if t is None and allow_none:
        return None
    if t.dim() != 1 or not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous 1-D tensor")
    if t.numel() < min_len:
        raise RuntimeError(f"{name}: needs at least {min_len} elements, has {t.numel()}")
    if t.numel() == 0:
        # an empty tensor has a NULL data pointer, which the C ABI rejects as "missing":
        # hand it a valid (never dereferenced) address instead
        _p(t, dtype, name)
        key = (t.device, dtype)
        if key not in _dummies:
            _dummies[key] = torch.zeros(8, dtype=dtype, device=t.device)
        return c_void_p(_dummies[key].data_ptr())
    return _p(t, dtype, name)


def _count(t, name):
    """Optional device-side count: a 0-d / 1-element int64 CUDA tensor."""
    if t is None:
        return None
    if t.numel() != 1:
        raise RuntimeError(f"{name}: device count must have one element")
    return _p(t, torch.int64, name)


# ------------------------------------------------------------------------------- a-1
def csc_build(edge_index, num_nodes):
    """to_csc (reference samplers/uniform_loader.py:32) -> (colptr, row, perm)."""
    lib = _lib.load()
    if edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise RuntimeError("csc_build: edge_index must be [2, E]")
    edge_index = edge_index.contiguous()
    E = edge_index.size(1)
    dev = edge_index.device
    if E > 0:
        lo, hi = int(edge_index.min()), int(edge_index.max())
        if lo < 0 or hi >= num_nodes:
            raise RuntimeError(f"csc_build: node id out of range [0,{num_nodes}): min {lo} max {hi}")
    colptr = torch.empty(num_nodes + 1, dtype=torch.int64, device=dev)
    row = torch.empty(E, dtype=torch.int64, device=dev)
    perm = torch.empty(E, dtype=torch.int64, device=dev)
    ws_bytes = lib.sage_csc_build_workspace_bytes(E, num_nodes)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    rc = lib.sage_csc_build(_p(edge_index[0], torch.int64, "edge_row"),
                            _p(edge_index[1], torch.int64, "edge_col"), E, num_nodes,
                            _p(colptr, torch.int64, "colptr"), _p(row, torch.int64, "row"),
                            _p(perm, torch.int64, "perm"), c_void_p(ws.data_ptr()), ws_bytes,
                            _stream())
    _lib.check(rc, "sage_csc_build")
    return colptr, row, perm


# ------------------------------------------------------------------------------- a-2
class SampleBuffers:
    """Caller-owned output/workspace buffers of sage_sample_fanout, reusable across calls."""

    def __init__(self, num_nodes, node_cap, edge_cap, device):
        lib = _lib.load()
        self.num_nodes, self.node_cap, self.edge_cap = int(num_nodes), int(node_cap), int(edge_cap)
        i64 = dict(dtype=torch.int64, device=device)
        self.node = torch.empty(self.node_cap, **i64)
        self.src = torch.empty(max(self.edge_cap, 1), **i64)
        self.dst = torch.empty(max(self.edge_cap, 1), **i64)
        self.eid = torch.empty(max(self.edge_cap, 1), **i64)
        self.seg_ptr = torch.empty(self.node_cap + 1, **i64)
        self.state = torch.zeros(STATE_WORDS, **i64)
        self.ws_bytes = lib.sage_sample_workspace_bytes(self.node_cap, self.edge_cap)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)


def new_relabel_table(num_nodes, device):
    """All-ones 64-bit words (see sage_hip.h: relabel_table)."""
    return torch.full((num_nodes,), -1, dtype=torch.int64, device=device)


def sample_fanout(colptr, row, seeds, fanout, rng_seed, rng_counter, table, stamp, buf,
                  ctrl=None, num_seeds=None):
    """torch_sparse.neighbor_sample (reference samplers/uniform_loader.py:64-72).

    Enqueues the sampler; results land in `buf` (device), sizes in buf.state.  No sync.
    ctrl: optional device control block (int64[CTRL_WORDS]) added to (rng_counter, stamp) and
    giving the offset of the batch inside `seeds` — see sage_sample_fanout_ctrl; the caller
    guarantees offset + num_seeds <= len(seeds).
    """
    lib = _lib.load()
    num_nodes = colptr.numel() - 1
    if buf.num_nodes != num_nodes or table.numel() != num_nodes:
        raise RuntimeError("sample_fanout: buffers/table were sized for another graph")
    hops = len(fanout)
    fan = (ctypes.c_int * hops)(*[int(k) for k in fanout])
    B = seeds.numel() if num_seeds is None else int(num_seeds)
    if B > seeds.numel() or B > buf.node_cap:
        raise RuntimeError("sample_fanout: num_seeds exceeds the seed buffer / node capacity")
    if ctrl is not None and ctrl.numel() < CTRL_WORDS:
        raise RuntimeError("sample_fanout: control block too small")
    rc = lib.sage_sample_fanout_ctrl(
        _vec(colptr, torch.int64, "colptr"), _vec(row, torch.int64, "row"), num_nodes,
        _vec(seeds, torch.int64, "seeds"), B, fan, hops,
        int(rng_seed) & 0xFFFFFFFFFFFFFFFF, int(rng_counter) & 0xFFFFFFFFFFFFFFFF,
        _p(table, torch.int64, "relabel_table"), int(stamp),
        _vec(ctrl, torch.int64, "ctrl", allow_none=True), buf.node_cap, buf.edge_cap,
        _p(buf.node, torch.int64, "node"), _p(buf.src, torch.int64, "src"),
        _p(buf.dst, torch.int64, "dst"), _p(buf.eid, torch.int64, "eid"),
        _p(buf.seg_ptr, torch.int64, "seg_ptr"), _p(buf.state, torch.int64, "state"),
        c_void_p(buf.ws.data_ptr()), buf.ws_bytes, _stream())
    _lib.check(rc, "sage_sample_fanout")


STATIC_SCRATCH_BYTES = 8 << 20


def sample_fanout_static(colptr, row, seeds, fanout, rng_seed, rng_counter, table, stamp, buf):
    """sample_fanout through sage_sample_fanout_static: the same results with the library's own
    static device array as the workspace (buf.ws is not used) — the entry point for callers that
    cannot allocate.  Capacities beyond STATIC_SCRATCH_BYTES of workspace raise RuntimeError."""
    lib = _lib.load()
    num_nodes = colptr.numel() - 1
    if buf.num_nodes != num_nodes or table.numel() != num_nodes:
        raise RuntimeError("sample_fanout_static: buffers/table were sized for another graph")
    hops = len(fanout)
    fan = (ctypes.c_int * hops)(*[int(k) for k in fanout])
    B = seeds.numel()
    if B > buf.node_cap:
        raise RuntimeError("sample_fanout_static: num_seeds exceeds the node capacity")
    rc = lib.sage_sample_fanout_static(
        _vec(colptr, torch.int64, "colptr"), _vec(row, torch.int64, "row"), num_nodes,
        _vec(seeds, torch.int64, "seeds"), B, fan, hops,
        int(rng_seed) & 0xFFFFFFFFFFFFFFFF, int(rng_counter) & 0xFFFFFFFFFFFFFFFF,
        _p(table, torch.int64, "relabel_table"), int(stamp), buf.node_cap, buf.edge_cap,
        _p(buf.node, torch.int64, "node"), _p(buf.src, torch.int64, "src"),
        _p(buf.dst, torch.int64, "dst"), _p(buf.eid, torch.int64, "eid"),
        _p(buf.seg_ptr, torch.int64, "seg_ptr"), _p(buf.state, torch.int64, "state"), _stream())
    _lib.check(rc, "sage_sample_fanout_static")


def state_accumulate(state, total):
    """total += state (both int64[STATE_WORDS] on the device)."""
    lib = _lib.load()
    rc = lib.sage_state_accumulate(_vec(state, torch.int64, "state", min_len=STATE_WORDS),
                                   _vec(total, torch.int64, "total", min_len=STATE_WORDS), _stream())
    _lib.check(rc, "sage_state_accumulate")


def ctrl_advance(ctrl, d_counter, d_stamp, d_offset, offset_wrap):
    lib = _lib.load()
    rc = lib.sage_ctrl_advance(_vec(ctrl, torch.int64, "ctrl", min_len=CTRL_WORDS), int(d_counter),
                               int(d_stamp), int(d_offset), int(offset_wrap), _stream())
    _lib.check(rc, "sage_ctrl_advance")


def random_neighbor(colptr, row, start, rng_seed, rng_counter, out=None, checked=True, num_dev=None):
    """torch_cluster.random_walk(..., walk_length=1)[:, 1] (reference samplers/utils.py:22).
    checked=False skips the host-side range check of `start` (it synchronises; the kernel itself
    never dereferences an id outside the graph).  num_dev: optional device-side count limiting
    how many entries of `start` are processed (the sampler's node count)."""
    lib = _lib.load()
    start = start.contiguous()
    if out is None:
        out = torch.empty_like(start)
    elif out.numel() < start.numel() or not out.is_contiguous():
        raise RuntimeError("random_neighbor: bad output buffer")
    if checked and start.numel() and (int(start.min()) < 0 or int(start.max()) >= colptr.numel() - 1):
        raise RuntimeError("random_neighbor: start node out of range")
    rc = lib.sage_random_neighbor(_vec(colptr, torch.int64, "colptr"), _vec(row, torch.int64, "row"),
                                  colptr.numel() - 1, _vec(start, torch.int64, "start"), start.numel(),
                                  _count(num_dev, "num_dev"),
                                  int(rng_seed) & 0xFFFFFFFFFFFFFFFF,
                                  int(rng_counter) & 0xFFFFFFFFFFFFFFFF,
                                  _p(out, torch.int64, "out"), _stream())
    _lib.check(rc, "sage_random_neighbor")
    return out


def random_nodes(num_nodes, n, rng_seed, rng_counter, device=None, out=None):
    """torch.randint(0, num_nodes, (n,)) on a Philox counter stream (reference
    samplers/utils.py:30-31: the negative samples)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(n, dtype=torch.int64, device=device)
    elif out.numel() < n or not out.is_contiguous():
        raise RuntimeError("random_nodes: bad output buffer")
    rc = lib.sage_random_nodes(int(num_nodes), int(n), int(rng_seed) & 0xFFFFFFFFFFFFFFFF,
                               int(rng_counter) & 0xFFFFFFFFFFFFFFFF, _p(out, torch.int64, "out"), _stream())
    _lib.check(rc, "sage_random_nodes")
    return out


def unsup_loss(out, pos, neg, d_out, d_pos, d_neg, row_loss, loss, m=None, grad_scale=1.0):
    """Unsupervised loss (reference trainers/base_trainers.py:179-181) + gradients; d_out is
    accumulated, d_pos / d_neg overwritten.  All [m, n] fp32 with one common row stride per group."""
    lib = _lib.load()
    op, ld = _mat(out, "out")
    pp, ld_p = _mat(pos, "pos")
    np_, ld_n = _mat(neg, "neg")
    dop, ld_d = _mat(d_out, "d_out")
    dpp, ld_dp = _mat(d_pos, "d_pos")
    dnp, ld_dn = _mat(d_neg, "d_neg")
    m = out.size(0) if m is None else int(m)
    n = out.size(1)
    if ld_p != ld or ld_n != ld or ld_dp != ld_d or ld_dn != ld_d:
        raise RuntimeError("unsup_loss: out/pos/neg (and the three gradients) must share a row stride")
    for t in (out, pos, neg, d_out, d_pos, d_neg):
        if t.size(0) < m or t.size(1) != n:
            raise RuntimeError("unsup_loss: shape mismatch")
    if row_loss.numel() < m or m <= 0:
        raise RuntimeError("unsup_loss: bad row_loss / m")
    rc = lib.sage_unsup_loss(op, pp, np_, ld, m, n, float(grad_scale), dop, dpp, dnp, ld_d,
                             _vec(row_loss, torch.float32, "row_loss"), _p(loss, torch.float32, "loss"), _stream())
    _lib.check(rc, "sage_unsup_loss")


# ------------------------------------------------------------------------------- a-3
def gather_rows(table, index, num_rows=None, num_rows_dev=None, out=None, checked=False):
    """out[i] = table[index[i]] for fp32 [N,F] tables and int64 [N] / [N,W] tables
    (filter_data's index_select, reference samplers/uniform_loader.py:238-239)."""
    lib = _lib.load()
    if table.dtype == torch.float32:
        words = 1
    elif table.dtype == torch.int64:
        words = 2
    else:
        raise RuntimeError(f"gather_rows: unsupported dtype {table.dtype}")
    squeeze = table.dim() == 1
    t2 = table.unsqueeze(1) if squeeze else table
    if t2.dim() != 2 or (t2.size(1) > 1 and t2.stride(1) != 1) or not t2.is_cuda:
        raise RuntimeError("gather_rows: table must be a CUDA tensor with contiguous rows")
    n = index.numel() if num_rows is None else int(num_rows)
    if n > index.numel():
        raise RuntimeError("gather_rows: num_rows exceeds len(index)")
    if not checked and n > 0:
        lo, hi = int(index[:n].min()), int(index[:n].max())
        if lo < 0 or hi >= t2.size(0):
            raise RuntimeError(f"gather_rows: index out of range [0,{t2.size(0)}): min {lo} max {hi}")
    width = t2.size(1) * words
    ld_table = (t2.stride(0) if t2.size(0) > 1 else t2.size(1)) * words
    if out is None:
        out = empty_rows(n, t2.size(1), table.device, table.dtype) if table.dim() == 2 else \
            torch.empty((n, t2.size(1)), dtype=table.dtype, device=table.device)
    o2 = out.unsqueeze(1) if out.dim() == 1 else out
    if o2.size(0) < n or o2.size(1) != t2.size(1) or o2.dtype != table.dtype or \
            (o2.size(1) > 1 and o2.stride(1) != 1):
        raise RuntimeError("gather_rows: bad output buffer")
    ld_out = (o2.stride(0) if o2.size(0) > 1 else o2.size(1)) * words
    rc = lib.sage_gather_rows(c_void_p(t2.data_ptr()), ld_table, _vec(index, torch.int64, "index"),
                              n, _count(num_rows_dev, "num_rows_dev"), width,
                              c_void_p(o2.data_ptr()), ld_out, _stream())
    _lib.check(rc, "sage_gather_rows")
    if squeeze and out.dim() == 2:
        out = out.squeeze(1)
    return out


# ------------------------------------------------------------------------------- a-5
def gather_reduce_fwd(x, seg_ptr, src, num_targets, reduce, node_map=None,
                      num_targets_dev=None, out=None, want_argmax=None, argmax=None, events=None):
    """scatter(x[src], dst, reduce) over target segments, fused
    (reference layers/sage.py:136-137 + :209-210).  Returns (out, argmax or None).
    events = (start, stop) raw HIP event handles: the launch carries them, so that their
    elapsed time is this kernel's own duration (measurement only)."""
    lib = _lib.load()
    xp, ld_x = _mat(x, "x")
    F = x.size(1)
    T = int(num_targets)
    if seg_ptr.numel() < T + 1:
        raise RuntimeError("gather_reduce_fwd: seg_ptr shorter than num_targets+1")
    if out is None:
        out = empty_rows(T, F, x.device)
    op, ld_out = _mat(out, "out")
    if out.size(0) < T or out.size(1) != F:
        raise RuntimeError("gather_reduce_fwd: bad output buffer")
    red = REDUCE[reduce]
    if red == 3 and (node_map.numel() if node_map is not None else x.size(0)) < T:
        raise RuntimeError("gather_reduce_fwd: gcn reads the targets' own rows: they must be rows of x (node_map)")
    if want_argmax is None:
        want_argmax = red == 2
    if red == 2 and want_argmax:
        if argmax is None:
            argmax = torch.empty((T, F), dtype=torch.int64, device=x.device)
        elif argmax.shape != (T, F) or not argmax.is_contiguous():
            raise RuntimeError("gather_reduce_fwd: argmax buffer must be a contiguous [T,F] int64 tensor")
    else:
        argmax = None
    head = (xp, ld_x, _vec(node_map, torch.int64, "node_map", allow_none=True),
            _vec(seg_ptr, torch.int64, "seg_ptr"), _vec(src, torch.int64, "src"), T,
            _count(num_targets_dev, "num_targets_dev"), F, red, op, ld_out,
            _p(argmax, torch.int64, "argmax", allow_none=True))
    if events is None:
        _lib.check(lib.sage_gather_reduce_fwd(*head, _stream()), "sage_gather_reduce_fwd")
    else:
        # raw hipEvent_t handles (e.g. torch.cuda.Event.cuda_event of events that exist already)
        e0, e1 = (int(e) for e in events)
        if not e0 or not e1:
            raise RuntimeError("gather_reduce_fwd: events must be two created HIP events")
        _lib.check(lib.sage_gather_reduce_fwd_timed(*head, c_void_p(e0), c_void_p(e1), _stream()),
                   "sage_gather_reduce_fwd_timed")
    return out, argmax


def gather_reduce_bwd(dout, seg_ptr, src, num_targets, reduce, dx, argmax=None,
                      num_targets_dev=None):
    """Accumulates the gradient of gather_reduce_fwd w.r.t. x into dx (pre-initialised)."""
    lib = _lib.load()
    dp, ld_dout = _mat(dout, "dout")
    xp, ld_dx = _mat(dx, "dx")
    F = dout.size(1)
    T = int(num_targets)
    if dx.size(1) != F or dout.size(0) < T or seg_ptr.numel() < T + 1:
        raise RuntimeError("gather_reduce_bwd: shape mismatch")
    red = REDUCE[reduce]
    if red == 2 and (argmax is None or argmax.shape != (T, F) or not argmax.is_contiguous()):
        raise RuntimeError("gather_reduce_bwd: max needs the forward's argmax [T,F]")
    rc = lib.sage_gather_reduce_bwd(dp, ld_dout, _vec(seg_ptr, torch.int64, "seg_ptr"),
                                    _vec(src, torch.int64, "src"),
                                    _p(argmax, torch.int64, "argmax", allow_none=True), T,
                                    _count(num_targets_dev, "num_targets_dev"), F, red, xp, ld_dx,
                                    _stream())
    _lib.check(rc, "sage_gather_reduce_bwd")
    return dx


def coo_to_csr(edge_src, edge_dst, num_targets):
    """Stable sort of an arbitrary edge list by target -> (seg_ptr, src_sorted, perm)."""
    lib = _lib.load()
    edge_src = edge_src.contiguous()
    edge_dst = edge_dst.contiguous()
    E = edge_src.numel()
    T = int(num_targets)
    dev = edge_src.device
    if E > 0:
        lo, hi = int(edge_dst.min()), int(edge_dst.max())
        if lo < 0 or hi >= T:
            raise RuntimeError(f"coo_to_csr: target id out of range [0,{T}): min {lo} max {hi}")
    seg_ptr = torch.empty(T + 1, dtype=torch.int64, device=dev)
    src_sorted = torch.empty(E, dtype=torch.int64, device=dev)
    perm = torch.empty(E, dtype=torch.int64, device=dev)
    ws_bytes = lib.sage_coo_to_csr_workspace_bytes(E, T)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    rc = lib.sage_coo_to_csr(_vec(edge_src, torch.int64, "edge_src"),
                             _vec(edge_dst, torch.int64, "edge_dst"), E, T,
                             _p(seg_ptr, torch.int64, "seg_ptr"),
                             _p(src_sorted, torch.int64, "src_sorted"),
                             _p(perm, torch.int64, "perm"), c_void_p(ws.data_ptr()), ws_bytes,
                             _stream())
    _lib.check(rc, "sage_coo_to_csr")
    return seg_ptr, src_sorted, perm


# ------------------------------------------------------------------------------- a-4
def _rows(t, m, name):
    if t is None:
        return None
    if t.numel() < m:
        raise RuntimeError(f"{name}: row map shorter than m")
    return _vec(t, torch.int64, name)


def linear_concat_fwd(a1, w1, a2=None, w2=None, bias=None, relu=False, m=None, m_dev=None,
                      rows1=None, rows2=None, out=None, accumulate=False):
    """out = a1[rows1] @ w1^T + a2[rows2] @ w2^T (+bias)(relu)
    (reference layers/sage.py:124-129; pool Linear :76,:202).  accumulate=True: out += product
    (out must be given and hold the addend; excludes bias / relu); accumulate='zeroed': the same
    with the caller's promise that the addend is all zeros (SAGE_EPI_ZEROED: shapes whose output
    elements have one owner are then stored, not load-add-stored)."""
    lib = _lib.load()
    a1p, ld_a1 = _mat(a1, "a1")
    w1p, ld_w1 = _mat(w1, "w1")
    n, k1 = w1.shape
    if a1.size(1) != k1:
        raise RuntimeError("linear_concat_fwd: a1/w1 inner dimensions differ")
    if m is None:
        m = rows1.numel() if rows1 is not None else a1.size(0)
    m = int(m)
    if rows1 is None and a1.size(0) < m:
        raise RuntimeError("linear_concat_fwd: a1 has fewer than m rows")
    k2 = 0
    a2p = w2p = None
    ld_a2 = ld_w2 = 0
    if a2 is not None:
        a2p, ld_a2 = _mat(a2, "a2")
        w2p, ld_w2 = _mat(w2, "w2")
        k2 = w2.size(1)
        if a2.size(1) != k2 or w2.size(0) != n:
            raise RuntimeError("linear_concat_fwd: a2/w2 shape mismatch")
        if rows2 is None and a2.size(0) < m:
            raise RuntimeError("linear_concat_fwd: a2 has fewer than m rows")
    if out is None:
        out = empty_rows(m, n, a1.device)
    op, ld_out = _mat(out, "out")
    if out.size(0) < m or out.size(1) != n:
        raise RuntimeError("linear_concat_fwd: bad output buffer")
    epi = (EPI_BIAS if bias is not None else 0) | (EPI_RELU if relu else 0)
    if accumulate:
        if epi:
            raise RuntimeError("linear_concat_fwd: accumulate excludes bias / relu")
        epi = EPI_ACCUM | (EPI_ZEROED if accumulate == 'zeroed' else 0)
    if bias is not None and bias.numel() != n:
        raise RuntimeError("linear_concat_fwd: bias length != n")
    rc = lib.sage_linear_concat_fwd(a1p, ld_a1, _rows(rows1, m, "rows1"), k1, w1p, ld_w1,
                                    a2p, ld_a2, _rows(rows2, m, "rows2"), k2, w2p, ld_w2,
                                    _vec(bias, torch.float32, "bias", allow_none=True), epi, m,
                                    _count(m_dev, "m_dev"), n, op, ld_out, _stream())
    _lib.check(rc, "sage_linear_concat_fwd")
    return out


def linear_dinput(dout, w, m=None, m_dev=None, out=None, accumulate=False):
    """da = dout @ w  (w is [n, k])."""
    lib = _lib.load()
    dp, ld_dout = _mat(dout, "dout")
    wp, ld_w = _mat(w, "w")
    n, k = w.shape
    m = dout.size(0) if m is None else int(m)
    if dout.size(1) != n or dout.size(0) < m:
        raise RuntimeError("linear_dinput: shape mismatch")
    if out is None:
        if accumulate:
            raise RuntimeError("linear_dinput: accumulate needs an output buffer")
        out = empty_rows(m, k, dout.device)
    op, ld_da = _mat(out, "da")
    if out.size(0) < m or out.size(1) != k:
        raise RuntimeError("linear_dinput: bad output buffer")
    rc = lib.sage_linear_dinput(dp, ld_dout, wp, ld_w, m, _count(m_dev, "m_dev"), n, k, op, ld_da,
                                1 if accumulate else 0, _stream())
    _lib.check(rc, "sage_linear_dinput")
    return out


def linear_dinput2(dout, w1, out1, w2, out2, m=None, m_dev=None):
    """out1 = dout @ w1, out2 = dout @ w2 in one launch (both overwritten)."""
    lib = _lib.load()
    dp, ld_d = _mat(dout, "dout")
    w1p, ld_w1 = _mat(w1, "w1")
    w2p, ld_w2 = _mat(w2, "w2")
    o1p, ld_o1 = _mat(out1, "out1")
    o2p, ld_o2 = _mat(out2, "out2")
    n, k = w1.shape
    m = dout.size(0) if m is None else int(m)
    if tuple(w2.shape) != (n, k) or dout.size(1) != n or dout.size(0) < m or \
            out1.size(0) < m or out2.size(0) < m or out1.size(1) != k or out2.size(1) != k:
        raise RuntimeError("linear_dinput2: shape mismatch")
    rc = lib.sage_linear_dinput2(dp, ld_d, w1p, ld_w1, o1p, ld_o1, w2p, ld_w2, o2p, ld_o2, m,
                                 _count(m_dev, "m_dev"), n, k, _stream())
    _lib.check(rc, "sage_linear_dinput2")


def linear_dweight(dout, a, m=None, m_dev=None, rows=None, dw=None, dbias=None):
    """dw += dout^T @ a[rows]  (dw is [n, k]; zero-initialised if not given)."""
    lib = _lib.load()
    dp, ld_dout = _mat(dout, "dout")
    ap, ld_a = _mat(a, "a")
    n, k = dout.size(1), a.size(1)
    if m is None:
        m = rows.numel() if rows is not None else dout.size(0)
    m = int(m)
    if dout.size(0) < m or (rows is None and a.size(0) < m):
        raise RuntimeError("linear_dweight: fewer than m rows")
    if dw is None:
        dw = torch.zeros((n, k), dtype=torch.float32, device=dout.device)
    wp, ld_dw = _mat(dw, "dw")
    if dw.shape != (n, k):
        raise RuntimeError("linear_dweight: bad dw buffer")
    if dbias is not None and dbias.numel() != n:
        raise RuntimeError("linear_dweight: dbias length != n")
    rc = lib.sage_linear_dweight(dp, ld_dout, ap, ld_a, _rows(rows, m, "rows"), m,
                                 _count(m_dev, "m_dev"), n, k, wp, ld_dw,
                                 _vec(dbias, torch.float32, "dbias", allow_none=True), _stream())
    _lib.check(rc, "sage_linear_dweight")
    return dw


def linear_dweight2(dout, a1, dw1, a2, dw2, m=None, m_dev=None, rows1=None, rows2=None):
    """dw1 += dout^T @ a1[rows1]; dw2 += dout^T @ a2[rows2] in one launch."""
    lib = _lib.load()
    dp, ld_dout = _mat(dout, "dout")
    a1p, ld_a1 = _mat(a1, "a1")
    a2p, ld_a2 = _mat(a2, "a2")
    w1p, ld_dw1 = _mat(dw1, "dw1")
    w2p, ld_dw2 = _mat(dw2, "dw2")
    n, k1, k2 = dout.size(1), a1.size(1), a2.size(1)
    if m is None:
        m = dout.size(0)
    m = int(m)
    if dout.size(0) < m or (rows1 is None and a1.size(0) < m) or (rows2 is None and a2.size(0) < m) or \
            dw1.shape != (n, k1) or dw2.shape != (n, k2):
        raise RuntimeError("linear_dweight2: shape mismatch")
    rc = lib.sage_linear_dweight2(dp, ld_dout, a1p, ld_a1, _rows(rows1, m, "rows1"), k1, w1p, ld_dw1,
                                  a2p, ld_a2, _rows(rows2, m, "rows2"), k2, w2p, ld_dw2, m,
                                  _count(m_dev, "m_dev"), n, _stream())
    _lib.check(rc, "sage_linear_dweight2")


def linear_dweight2_parts(dout, a1, dw1, a2, dw2, part_stride, max_parts, holder, m=None, m_dev=None, rows1=None,
                          rows2=None):
    """linear_dweight2 whose split-K slices STORE their partial sums at dw1 / dw2 + z * part_stride
    floats where the launch allows it; returns the number of parts (1: the sums were added into
    dw1 / dw2 as usual).  The caller owns max_parts * part_stride floats behind dw1's storage, and
    `holder`, a ctypes.c_int that outlives every replay of the recorded call (the C side writes the
    number of parts through its address)."""
    lib = _lib.load()
    dp, ld_dout = _mat(dout, "dout")
    a1p, ld_a1 = _mat(a1, "a1")
    a2p, ld_a2 = _mat(a2, "a2")
    w1p, ld_dw1 = _mat(dw1, "dw1")
    w2p, ld_dw2 = _mat(dw2, "dw2")
    n, k1, k2 = dout.size(1), a1.size(1), a2.size(1)
    if m is None:
        m = dout.size(0)
    m = int(m)
    if dout.size(0) < m or (rows1 is None and a1.size(0) < m) or (rows2 is None and a2.size(0) < m) or \
            dw1.shape != (n, k1) or dw2.shape != (n, k2) or part_stride <= 0 or max_parts < 1:
        raise RuntimeError("linear_dweight2_parts: shape mismatch")
    if not isinstance(holder, ctypes.c_int):
        raise RuntimeError("linear_dweight2_parts: holder must be a ctypes.c_int kept alive by the caller")
    parts = holder
    rc = lib.sage_linear_dweight2_parts(dp, ld_dout, a1p, ld_a1, _rows(rows1, m, "rows1"), k1, w1p, ld_dw1,
                                        a2p, ld_a2, _rows(rows2, m, "rows2"), k2, w2p, ld_dw2, m,
                                        _count(m_dev, "m_dev"), n, int(part_stride), int(max_parts),
                                        ctypes.addressof(parts), _stream())
    _lib.check(rc, "sage_linear_dweight2_parts")
    return parts.value


# ------------------------------------------------------------------------------- rows
def l2norm_relu_fwd(z, relu, m=None, m_dev=None, y=None, h=None, inv_norm=None, clear_z=False):
    """y = normalize(z); h = relu(y) if relu (reference layers/sage.py:131-132,
    models/graphsage.py:47-48).  Returns (y, h or None, inv_norm).  clear_z=True zeroes the
    consumed rows of z (its producer may then accumulate into it)."""
    lib = _lib.load()
    zp, ld_z = _mat(z, "z")
    m = z.size(0) if m is None else int(m)
    n = z.size(1)
    if y is None:
        y = torch.empty((m, n), dtype=torch.float32, device=z.device)
    if relu and h is None:
        h = torch.empty((m, n), dtype=torch.float32, device=z.device)
    if inv_norm is None:
        inv_norm = torch.empty(m, dtype=torch.float32, device=z.device)
    yp, ld_y = _mat(y, "y")
    hp, ld_h = (None, 0)
    if relu:
        hp, ld_h = _mat(h, "h")
        if h.size(0) < m or h.size(1) != n:
            raise RuntimeError("l2norm_relu_fwd: bad h buffer")
    if z.size(0) < m or y.size(0) < m or y.size(1) != n or inv_norm.numel() < m:
        raise RuntimeError("l2norm_relu_fwd: bad buffers")
    rc = lib.sage_l2norm_relu_fwd(zp, ld_z, m, _count(m_dev, "m_dev"), n, yp, ld_y, hp, ld_h,
                                  _vec(inv_norm, torch.float32, "inv_norm"), int(bool(clear_z)), _stream())
    _lib.check(rc, "sage_l2norm_relu_fwd")
    return y, (h if relu else None), inv_norm


def relu_bwd(dout, out, m=None, m_dev=None, dz=None, clear_dout=False):
    """dz = dout * [out > 0] (threshold_backward of relu(pool(x)), reference layers/sage.py:202,206).
    clear_dout=True zeroes the consumed rows of dout."""
    lib = _lib.load()
    gp, ld_g = _mat(dout, "dout")
    op, ld_o = _mat(out, "out")
    m = dout.size(0) if m is None else int(m)
    n = dout.size(1)
    if out.size(1) != n or out.size(0) < m or dout.size(0) < m:
        raise RuntimeError("relu_bwd: shape mismatch")
    if dz is None:
        dz = empty_rows(m, n, dout.device)
    zp, ld_z = _mat(dz, "dz")
    if dz.size(0) < m or dz.size(1) != n:
        raise RuntimeError("relu_bwd: bad output buffer")
    rc = lib.sage_relu_bwd(gp, ld_g, op, ld_o, m, _count(m_dev, "m_dev"), n, zp, ld_z, 1 if clear_dout else 0,
                           _stream())
    _lib.check(rc, "sage_relu_bwd")
    return dz


def pool_linear_fwd(x, w, bias, m=None, m_dev=None, rows=None, out=None, dpre=None):
    """out = relu(x[rows] @ w^T + bias) (reference layers/sage.py:76,202,206); dpre (out's shape):
    its rows < m come back zeroed from the same launch."""
    lib = _lib.load()
    xp, ld_x = _mat(x, "x")
    wp, ld_w = _mat(w, "w")
    n, k = w.shape
    if x.size(1) != k or bias.numel() != n:
        raise RuntimeError("pool_linear_fwd: x / w / bias shapes differ")
    if m is None:
        m = rows.numel() if rows is not None else x.size(0)
    m = int(m)
    if rows is None and x.size(0) < m:
        raise RuntimeError("pool_linear_fwd: x has fewer than m rows")
    if out is None:
        out = empty_rows(m, n, x.device)
    op, ld_o = _mat(out, "out")
    if out.size(0) < m or out.size(1) != n:
        raise RuntimeError("pool_linear_fwd: bad output buffer")
    zp, ld_z = None, 0
    if dpre is not None:
        zp, ld_z = _mat(dpre, "dpre")
        if dpre.size(0) < m or dpre.size(1) != n:
            raise RuntimeError("pool_linear_fwd: bad dpre buffer")
    rc = lib.sage_pool_linear_fwd(xp, ld_x, _rows(rows, m, "rows"), k, wp, ld_w, _vec(bias, torch.float32, "bias"),
                                  m, _count(m_dev, "m_dev"), n, op, ld_o, zp, ld_z, _stream())
    _lib.check(rc, "sage_pool_linear_fwd")
    return out


def pool_max_bwd(dagg, agg, seg_ptr, src, argmax, num_targets, dpre, dbias=None, num_targets_dev=None,
                 indeg=None):
    """dpre[src[argmax[t, f]], f] += dagg[t, f] where agg[t, f] > 0; dbias += the column sums of
    what was routed (backward of max over relu(pool(x)) rows, reference layers/sage.py:201-207).
    dpre holds zeros on entry.  indeg: int32 workspace, one entry per row of dpre — rows with a
    single incoming edge are then written whole instead of added into."""
    lib = _lib.load()
    gp, ld_g = _mat(dagg, "dagg")
    ap, ld_a = _mat(agg, "agg")
    zp, ld_z = _mat(dpre, "dpre")
    T, F = int(num_targets), dagg.size(1)
    if agg.size(1) != F or dpre.size(1) != F or dagg.size(0) < T or agg.size(0) < T or seg_ptr.numel() < T + 1:
        raise RuntimeError("pool_max_bwd: shape mismatch")
    if argmax.shape != (T, F) or not argmax.is_contiguous():
        raise RuntimeError("pool_max_bwd: needs the forward's argmax [T,F]")
    if dbias is not None and dbias.numel() < F:
        raise RuntimeError("pool_max_bwd: dbias shorter than the feature count")
    S = dpre.size(0)
    if indeg is not None and (indeg.dtype != torch.int32 or indeg.numel() < S or not indeg.is_contiguous()):
        raise RuntimeError("pool_max_bwd: indeg must be a contiguous int32 tensor with one entry per dpre row")
    rc = lib.sage_pool_max_bwd(gp, ld_g, ap, ld_a, _vec(seg_ptr, torch.int64, "seg_ptr"),
                               _vec(src, torch.int64, "src"), _p(argmax, torch.int64, "argmax"),
                               T, _count(num_targets_dev, "num_targets_dev"), F, zp, ld_z, S,
                               _p(indeg, torch.int32, "indeg", allow_none=True),
                               _vec(dbias, torch.float32, "dbias", allow_none=True), _stream())
    _lib.check(rc, "sage_pool_max_bwd")
    return dpre


def pool_mean_bwd(dagg, p, seg_ptr, src, num_targets, dpre, num_targets_dev=None, indeg=None):
    """dpre[src[e], f] += dagg[t, f] / deg(t) where p[src[e], f] > 0, over every edge e of every
    target t (backward of mean over relu(pool(x)) rows, reference layers/sage.py:201-210).  dpre
    holds zeros on entry; indeg as in pool_max_bwd."""
    lib = _lib.load()
    gp, ld_g = _mat(dagg, "dagg")
    pp, ld_p = _mat(p, "p")
    zp, ld_z = _mat(dpre, "dpre")
    T, F = int(num_targets), dagg.size(1)
    S = dpre.size(0)
    if p.size(1) != F or dpre.size(1) != F or dagg.size(0) < T or p.size(0) < S or seg_ptr.numel() < T + 1:
        raise RuntimeError("pool_mean_bwd: shape mismatch")
    if indeg is not None and (indeg.dtype != torch.int32 or indeg.numel() < S or not indeg.is_contiguous()):
        raise RuntimeError("pool_mean_bwd: indeg must be a contiguous int32 tensor with one entry per dpre row")
    rc = lib.sage_pool_mean_bwd(gp, ld_g, pp, ld_p, _vec(seg_ptr, torch.int64, "seg_ptr"),
                                _vec(src, torch.int64, "src"), T, _count(num_targets_dev, "num_targets_dev"), F,
                                zp, ld_z, S, _p(indeg, torch.int32, "indeg", allow_none=True), _stream())
    _lib.check(rc, "sage_pool_mean_bwd")
    return dpre


class ReverseEdges:
    """The edges of a sampled hop listed per SOURCE row (sage_pool_reverse_edges): static buffers for
    up to `num_sources` rows and `edge_cap` edges, refilled by `build`."""

    def __init__(self, num_sources, edge_cap, device):
        lib = _lib.load()
        self.num_sources, self.edge_cap = int(num_sources), int(edge_cap)
        self.count = torch.zeros(self.num_sources + 1, dtype=torch.int32, device=device)
        self.ptr = torch.zeros(self.num_sources + 1, dtype=torch.int32, device=device)
        self.entries = torch.zeros(max(self.edge_cap, 1), dtype=torch.int32, device=device)
        self.ws_bytes = int(lib.sage_pool_reverse_edges_workspace_bytes(self.num_sources))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)

    def build(self, seg_ptr, src, num_targets, num_targets_dev=None):
        lib = _lib.load()
        T = int(num_targets)
        if seg_ptr.numel() < T + 1:
            raise RuntimeError("ReverseEdges.build: seg_ptr shorter than num_targets + 1")
        rc = lib.sage_pool_reverse_edges(_vec(seg_ptr, torch.int64, "seg_ptr"), _vec(src, torch.int64, "src"), T,
                                         _count(num_targets_dev, "num_targets_dev"), self.num_sources,
                                         _p(self.count, torch.int32, "count"), _p(self.ptr, torch.int32, "ptr"),
                                         _p(self.entries, torch.int32, "entries"), self.edge_cap,
                                         self.ws.data_ptr(), self.ws_bytes, _stream())
        _lib.check(rc, "sage_pool_reverse_edges")
        return self


def pool_max_fwd(p, seg_ptr, src, num_targets, max_segment, out, arg8, num_targets_dev=None):
    """out[t] = max over the rows p[src[e]] of target t's edges (p = relu(pool(x)), reference
    layers/sage.py:201-210) with the arg-max edge kept as one byte per (target, column) in arg8
    (uint8 [T, >= F rounded up to 4]): the edge's offset in the segment, 255 = no gradient."""
    lib = _lib.load()
    pp, ld_p = _mat(p, "p")
    op, ld_o = _mat(out, "out")
    T, F = int(num_targets), p.size(1)
    if (out.size(1) != F or out.size(0) < T or seg_ptr.numel() < T + 1 or arg8.dtype != torch.uint8 or arg8.dim() != 2
            or arg8.size(0) < T or arg8.stride(1) != 1):
        raise RuntimeError("pool_max_fwd: shape mismatch (arg8: uint8 [T, F padded to 4])")
    rc = lib.sage_pool_max_fwd(pp, ld_p, _vec(seg_ptr, torch.int64, "seg_ptr"), _vec(src, torch.int64, "src"), T,
                               _count(num_targets_dev, "num_targets_dev"), F, int(max_segment), op, ld_o,
                               arg8.data_ptr(), arg8.stride(0), _stream())
    _lib.check(rc, "sage_pool_max_fwd")
    return out


def pool_max_bwd_pull(dagg, arg8, rev, dpre, num_sources_dev=None):
    """pool_max_bwd pulled by the source rows: dpre[s] = sum over the in-edges (t, j) of row s of
    dagg[t, f] where arg8[t, f] == j; rows [0, rev.num_sources) of dpre are stored whole."""
    lib = _lib.load()
    gp, ld_g = _mat(dagg, "dagg")
    zp, ld_z = _mat(dpre, "dpre")
    F = dagg.size(1)
    if dpre.size(1) != F or dpre.size(0) < rev.num_sources or arg8.dtype != torch.uint8 or arg8.stride(1) != 1:
        raise RuntimeError("pool_max_bwd_pull: shape mismatch")
    rc = lib.sage_pool_max_bwd_pull(gp, ld_g, arg8.data_ptr(), arg8.stride(0), _p(rev.ptr, torch.int32, "ptr"),
                                    _p(rev.entries, torch.int32, "entries"), rev.edge_cap, rev.num_sources,
                                    _count(num_sources_dev, "num_sources_dev"), F, zp, ld_z, _stream())
    _lib.check(rc, "sage_pool_max_bwd_pull")
    return dpre


def pool_mean_bwd_pull(dagg, p, seg_ptr, rev, dpre, num_sources_dev=None):
    """pool_mean_bwd pulled by the source rows: dpre[s] = (p[s] > 0) * sum over the in-edges of row
    s of dagg[t] / deg(t); rows [0, rev.num_sources) of dpre are stored whole."""
    lib = _lib.load()
    gp, ld_g = _mat(dagg, "dagg")
    pp, ld_p = _mat(p, "p")
    zp, ld_z = _mat(dpre, "dpre")
    F = dagg.size(1)
    if p.size(1) != F or dpre.size(1) != F or dpre.size(0) < rev.num_sources or p.size(0) < rev.num_sources:
        raise RuntimeError("pool_mean_bwd_pull: shape mismatch")
    rc = lib.sage_pool_mean_bwd_pull(gp, ld_g, pp, ld_p, _vec(seg_ptr, torch.int64, "seg_ptr"),
                                     _p(rev.ptr, torch.int32, "ptr"), _p(rev.entries, torch.int32, "entries"),
                                     rev.edge_cap, rev.num_sources, _count(num_sources_dev, "num_sources_dev"), F,
                                     zp, ld_z, _stream())
    _lib.check(rc, "sage_pool_mean_bwd_pull")
    return dpre


def l2norm_relu_bwd(dh, y, inv_norm, relu, m=None, m_dev=None, dz=None, clear_dh=False):
    """Backward of relu(normalize(z)).  clear_dh=True zeroes the consumed rows of dh."""
    lib = _lib.load()
    dp, ld_dh = _mat(dh, "dh")
    yp, ld_y = _mat(y, "y")
    m = y.size(0) if m is None else int(m)
    n = y.size(1)
    if dz is None:
        dz = torch.empty((m, n), dtype=torch.float32, device=y.device)
    zp, ld_dz = _mat(dz, "dz")
    if dh.size(0) < m or dh.size(1) != n or dz.size(0) < m or dz.size(1) != n or \
            inv_norm.numel() < m or y.size(0) < m:
        raise RuntimeError("l2norm_relu_bwd: bad buffers")
    rc = lib.sage_l2norm_relu_bwd(dp, ld_dh, yp, ld_y, _vec(inv_norm, torch.float32, "inv_norm"),
                                  1 if relu else 0, m, _count(m_dev, "m_dev"), n, zp, ld_dz,
                                  1 if clear_dh else 0, _stream())
    _lib.check(rc, "sage_l2norm_relu_bwd")
    return dz


def l2norm_relu_bwd_z(dh, z, relu, m=None, m_dev=None, dz=None, clear_dh=False):
    """Backward of relu(normalize(z)) from the pre-normalisation rows z themselves (no y /
    inv_norm kept by a forward pass: logit_layer_fwd(h_root=...)).  z is handed back zeroed;
    clear_dh=True zeroes the consumed rows of dh as well."""
    lib = _lib.load()
    dp, ld_dh = _mat(dh, "dh")
    zp, ld_z = _mat(z, "z")
    m = z.size(0) if m is None else int(m)
    n = z.size(1)
    if dz is None:
        dz = torch.empty((m, n), dtype=torch.float32, device=z.device)
    op, ld_dz = _mat(dz, "dz")
    if dh.size(0) < m or dh.size(1) != n or dz.size(0) < m or dz.size(1) != n or z.size(0) < m:
        raise RuntimeError("l2norm_relu_bwd_z: bad buffers")
    rc = lib.sage_l2norm_relu_bwd_z(dp, ld_dh, zp, ld_z, 1 if relu else 0, m, _count(m_dev, "m_dev"), n, op, ld_dz,
                                    1 if clear_dh else 0, _stream())
    _lib.check(rc, "sage_l2norm_relu_bwd_z")
    return dz


def l2norm_relu_bwd_z_dw(dh, z, relu, m, m_dev, dz, clear_dh, dz2, m2, m2_dev, agg2, dw_l, h_root2, dw_r,
                         row_loss=None, loss=None):
    """l2norm_relu_bwd_z with the weight gradients of the class-logit layer above riding along:
    dw_l += dz2[:m2]^T @ agg2[:m2], dw_r += dz2[:m2]^T @ h_root2[:m2], loss = mean(row_loss[:m2])."""
    lib = _lib.load()
    dp, ld_dh = _mat(dh, "dh")
    zp, ld_z = _mat(z, "z")
    m, m2 = int(m), int(m2)
    n = z.size(1)
    op, ld_dz = _mat(dz, "dz")
    d2p, ld_dz2 = _mat(dz2, "dz2")
    ap, ld_agg = _mat(agg2, "agg2")
    glp, ld_gl = _mat(dw_l, "dw_l")
    n2, h2 = dw_l.shape
    if (h_root2 is None) != (dw_r is None):
        raise RuntimeError("l2norm_relu_bwd_z_dw: h_root2 and dw_r go together (both None: no root weight)")
    rp, ld_hr = (None, 0) if h_root2 is None else _mat(h_root2, "h_root2")
    grp, ld_gr = (None, 0) if dw_r is None else _mat(dw_r, "dw_r")
    if dh.size(0) < m or dh.size(1) != n or dz.size(0) < m or dz.size(1) != n or z.size(0) < m \
            or dz2.size(0) < m2 or dz2.size(1) != n2 or agg2.size(0) < m2 or agg2.size(1) != h2 \
            or (h_root2 is not None and (h_root2.size(0) < m2 or h_root2.size(1) != h2 or tuple(dw_r.shape) != (n2, h2))) \
            or (row_loss is None) != (loss is None) or (row_loss is not None and row_loss.numel() < m2):
        raise RuntimeError("l2norm_relu_bwd_z_dw: bad buffers")
    rc = lib.sage_l2norm_relu_bwd_z_dw(dp, ld_dh, zp, ld_z, 1 if relu else 0, m, _count(m_dev, "m_dev"), n, op, ld_dz,
                                       1 if clear_dh else 0, d2p, ld_dz2, m2, _count(m2_dev, "m2_dev"), n2, h2,
                                       ap, ld_agg, glp, ld_gl, rp, ld_hr, grp, ld_gr,
                                       _vec(row_loss, torch.float32, "row_loss", allow_none=True),
                                       _p(loss, torch.float32, "loss", allow_none=True), _stream())
    _lib.check(rc, "sage_l2norm_relu_bwd_z_dw")
    return dz


def softmax_xent(logits, labels, label_rows=None, m=None, grad_scale=1.0, loss=None,
                 dlogits=None, want_grad=True, checked=False, row_loss=None):
    """Mean cross-entropy + gradient (reference fig2a/runners/graphsage_runners.py:43).
    row_loss (float32[>= m] scratch): the row-parallel kernel writes the per-row losses there."""
    lib = _lib.load()
    lp, ld = _mat(logits, "logits")
    m = logits.size(0) if m is None else int(m)
    n = logits.size(1)
    if logits.size(0) < m or m <= 0:
        raise RuntimeError("softmax_xent: bad m")
    if label_rows is not None:
        if label_rows.numel() < m:
            raise RuntimeError("softmax_xent: label_rows shorter than m")
        if not checked and (int(label_rows[:m].min()) < 0 or int(label_rows[:m].max()) >= labels.numel()):
            raise RuntimeError("softmax_xent: label_rows out of range")
    elif labels.numel() < m:
        raise RuntimeError("softmax_xent: labels shorter than m")
    if loss is None:
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    if want_grad and dlogits is None:
        dlogits = torch.empty((m, n), dtype=torch.float32, device=logits.device)
    gp, ld_g = (None, 0)
    if dlogits is not None:
        gp, ld_g = _mat(dlogits, "dlogits")
        if dlogits.size(0) < m or dlogits.size(1) != n:
            raise RuntimeError("softmax_xent: bad dlogits buffer")
    if row_loss is not None:      # one wave per row over many workgroups, fixed-order mean
        if row_loss.numel() < m:
            raise RuntimeError("softmax_xent: row_loss shorter than m")
        rc = lib.sage_softmax_xent_rows(lp, ld, _vec(labels, torch.int64, "labels"),
                                        _vec(label_rows, torch.int64, "label_rows", allow_none=True), m, n,
                                        float(grad_scale), _vec(row_loss, torch.float32, "row_loss"),
                                        _p(loss, torch.float32, "loss"), gp, ld_g, _stream())
        _lib.check(rc, "sage_softmax_xent_rows")
        return loss, dlogits
    rc = lib.sage_softmax_xent(lp, ld, _vec(labels, torch.int64, "labels"),
                               _vec(label_rows, torch.int64, "label_rows", allow_none=True), m, n,
                               float(grad_scale), _p(loss, torch.float32, "loss"), gp, ld_g,
                               _stream())
    _lib.check(rc, "sage_softmax_xent")
    return loss, dlogits


def logits_xent_fused(z, labels, label_rows, m, y, inv_norm, dz, row_loss, loss, grad_scale=1.0,
                      clear_z=False):
    """normalize + cross-entropy + gradient w.r.t. z in one pass (last layer of the step).
    All buffers are caller-provided (static-shape engine); label_rows must be in range.
    clear_z=True zeroes the consumed rows of z (for a producer that accumulates into it)."""
    lib = _lib.load()
    zp, ld_z = _mat(z, "z")
    n = z.size(1)
    m = int(m)
    dp, ld_dz = _mat(dz, "dz")
    yp, ld_y = (None, 0) if y is None else _mat(y, "y")
    if z.size(0) < m or dz.size(0) < m or dz.size(1) != n or row_loss.numel() < m or m <= 0 or \
            (y is not None and (y.size(0) < m or y.size(1) != n)) or \
            (inv_norm is not None and inv_norm.numel() < m) or \
            (label_rows is not None and label_rows.numel() < m) or \
            (label_rows is None and labels.numel() < m):
        raise RuntimeError("logits_xent_fused: bad buffers")
    rc = lib.sage_logits_xent_fused(zp, ld_z, _vec(labels, torch.int64, "labels"),
                                    _vec(label_rows, torch.int64, "label_rows", allow_none=True), m, n,
                                    float(grad_scale), yp, ld_y,
                                    _vec(inv_norm, torch.float32, "inv_norm", allow_none=True), dp, ld_dz,
                                    _vec(row_loss, torch.float32, "row_loss"), _p(loss, torch.float32, "loss"),
                                    1 if clear_z else 0, _stream())
    _lib.check(rc, "sage_logits_xent_fused")


def logits_bce_fused(z, targets, label_rows, m, y=None, inv_norm=None, dz=None, row_loss=None, loss=None,
                     row_stats=None, grad_scale=1.0, clear_z=False):
    """normalize + BCE-with-logits (mean over m x n elements) + gradient w.r.t. z in one pass —
    the multi-label last layer (reference examples/ppi_graphsage_supervised.py:30).  targets: fp32
    [N, n] of 0 / 1, read through label_rows.  dz=None: forward only; row_stats int64 [m, 3]:
    per-row (tp, fp, fn) of the prediction y > 0."""
    lib = _lib.load()
    zp, ld_z = _mat(z, "z")
    tp_, ld_t = _mat(targets, "targets")
    n = z.size(1)
    m = int(m)
    yp, ld_y = (None, 0) if y is None else _mat(y, "y")
    dp, ld_dz = (None, 0) if dz is None else _mat(dz, "dz")
    if z.size(0) < m or m <= 0 or targets.size(1) != n or (dz is not None and (dz.size(0) < m or dz.size(1) != n)) \
            or (y is not None and (y.size(0) < m or y.size(1) != n)) or (row_loss is not None and row_loss.numel() < m) \
            or (inv_norm is not None and inv_norm.numel() < m) or (loss is not None and row_loss is None) \
            or (label_rows is not None and label_rows.numel() < m) or (label_rows is None and targets.size(0) < m) \
            or (row_stats is not None and (row_stats.numel() < 3 * m or not row_stats.is_contiguous())):
        raise RuntimeError("logits_bce_fused: bad buffers")
    rc = lib.sage_logits_bce_fused(zp, ld_z, tp_, ld_t, _vec(label_rows, torch.int64, "label_rows", allow_none=True),
                                   m, n, float(grad_scale), yp, ld_y,
                                   _vec(inv_norm, torch.float32, "inv_norm", allow_none=True), dp, ld_dz,
                                   _vec(row_loss, torch.float32, "row_loss", allow_none=True),
                                   _p(loss, torch.float32, "loss", allow_none=True),
                                   _p(row_stats, torch.int64, "row_stats", allow_none=True),
                                   1 if clear_z else 0, _stream())
    _lib.check(rc, "sage_logits_bce_fused")


def logit_layer_supported(h, n):
    """Whether sage_logit_layer_fwd / _bwd take a [n, h] + [n, h] weight pair (n outputs, width h)."""
    return n <= 64 and h % 4 == 0 and h <= 1024 and n * 2 * h * 4 <= 160 * 1024


def logit_layer_fwd(h_rows, seg_ptr, src, m, reduce, w_l, w_r, labels, label_rows, agg, y, inv_norm, dz,
                    row_loss, grad_scale=1.0, h_root=None, dh=None, writers=None):
    """The last SAGE layer on the m seed rows + normalise + cross entropy + d loss / d z in one
    launch (reference layers/sage.py:99-134 + fig2a/runners/graphsage_runners.py:41-43); all
    buffers are the caller's (static-shape engine).  agg [m, h] is kept for the weight gradients.
    h_root [>= m, h]: h_rows holds the previous layer's PRE-activation rows; the kernel takes
    every row through relu(normalize(.)) itself and stores the seeds' activations there.
    dh [rows of h_rows, h]: the layer's input gradient is added to it in the same launch
    (sage_logit_layer_fwd_bwd = this + logit_layer_bwd without the loss mean); writers (int32, one
    per row of dh, from count_row_writers): rows with one writer are stored instead of added —
    dh must then hold zeros."""
    lib = _lib.load()
    hp, ld_h = _mat(h_rows, "h_rows")
    wlp, ld_wl = _mat(w_l, "w_l")
    wrp, ld_wr = _mat(w_r, "w_r")
    n, h = w_l.shape
    m = int(m)
    ap, ld_agg = _mat(agg, "agg")
    dp, ld_dz = _mat(dz, "dz")
    yp, ld_y = (None, 0) if y is None else _mat(y, "y")
    if h_rows.size(1) != h or tuple(w_r.shape) != (n, h) or m <= 0 or h_rows.size(0) < m or agg.size(0) < m \
            or agg.size(1) != h or dz.size(0) < m or dz.size(1) != n or row_loss.numel() < m \
            or seg_ptr.numel() < m + 1 or (y is not None and (y.size(0) < m or y.size(1) != n)) \
            or (inv_norm is not None and inv_norm.numel() < m) \
            or (label_rows is not None and label_rows.numel() < m) or (label_rows is None and labels.numel() < m):
        raise RuntimeError("logit_layer_fwd: bad buffers")
    rp, ld_root = (None, 0) if h_root is None else _mat(h_root, "h_root")
    if h_root is not None and (h_root.size(0) < m or h_root.size(1) != h):
        raise RuntimeError("logit_layer_fwd: bad h_root buffer")
    args = (hp, ld_h, _vec(seg_ptr, torch.int64, "seg_ptr"), _vec(src, torch.int64, "src"), m, h,
            REDUCE[reduce], wlp, ld_wl, wrp, ld_wr, n, _vec(labels, torch.int64, "labels"),
            _vec(label_rows, torch.int64, "label_rows", allow_none=True), float(grad_scale),
            ap, ld_agg, yp, ld_y, _vec(inv_norm, torch.float32, "inv_norm", allow_none=True),
            dp, ld_dz, _vec(row_loss, torch.float32, "row_loss"), rp, ld_root)
    if dh is None:
        _lib.check(lib.sage_logit_layer_fwd(*args, _stream()), "sage_logit_layer_fwd")
        return
    gp, ld_dh = _mat(dh, "dh")
    if dh.size(1) != h or dh.size(0) < h_rows.size(0):
        raise RuntimeError("logit_layer_fwd: dh must cover the rows of h_rows")
    if writers is not None and (writers.dtype != torch.int32 or writers.numel() < dh.size(0) or not writers.is_contiguous()):
        raise RuntimeError("logit_layer_fwd: writers must be a contiguous int32 tensor with one entry per row of dh")
    _lib.check(lib.sage_logit_layer_fwd_bwd(*args, gp, ld_dh, _p(writers, torch.int32, "writers", allow_none=True),
                                            _stream()), "sage_logit_layer_fwd_bwd")


def logit_layer_pool_fwd_bwd(h_rows, p_rows, seg_ptr, src, m, reduce, w_l, w_r, labels, label_rows, agg, y, inv_norm, dz,
                             row_loss, dh, dpre, grad_scale=1.0, writers=None):
    """logit_layer_fwd(dh=...) for the last layer of a pooling aggregator (reference layers/sage.py:201-210):
    neighbour rows from p_rows = relu(pool(h)) (indexed like h_rows), root row h_rows[t]; reduce 'mean' / 'max'.
    dh[:m] = dz . W_r (stored), dpre[src] += the neighbours' share through the relu (dpre holds zeros on entry)."""
    lib = _lib.load()
    hp, ld_h = _mat(h_rows, "h_rows")
    pp, ld_p = _mat(p_rows, "p_rows")
    wlp, ld_wl = _mat(w_l, "w_l")
    wrp, ld_wr = _mat(w_r, "w_r")
    n, h = w_l.shape
    m = int(m)
    ap, ld_agg = _mat(agg, "agg")
    dp, ld_dz = _mat(dz, "dz")
    yp, ld_y = (None, 0) if y is None else _mat(y, "y")
    gp, ld_dh = _mat(dh, "dh")
    zp, ld_dpre = _mat(dpre, "dpre")
    if h_rows.size(1) != h or p_rows.size(1) != h or tuple(w_r.shape) != (n, h) or m <= 0 or h_rows.size(0) < m \
            or agg.size(0) < m or agg.size(1) != h or dz.size(0) < m or dz.size(1) != n or row_loss.numel() < m \
            or seg_ptr.numel() < m + 1 or (y is not None and (y.size(0) < m or y.size(1) != n)) \
            or (inv_norm is not None and inv_norm.numel() < m) or dh.size(1) != h or dh.size(0) < m \
            or dpre.size(1) != h or dpre.size(0) < p_rows.size(0) \
            or (label_rows is not None and label_rows.numel() < m) or (label_rows is None and labels.numel() < m):
        raise RuntimeError("logit_layer_pool_fwd_bwd: bad buffers")
    if writers is not None and (writers.dtype != torch.int32 or writers.numel() < p_rows.size(0) or not writers.is_contiguous()):
        raise RuntimeError("logit_layer_pool_fwd_bwd: writers must be a contiguous int32 tensor with one entry per row of p_rows")
    rc = lib.sage_logit_layer_pool_fwd_bwd(hp, ld_h, pp, ld_p, _vec(seg_ptr, torch.int64, "seg_ptr"), _vec(src, torch.int64, "src"),
                                           m, h, REDUCE[reduce], wlp, ld_wl, wrp, ld_wr, n, _vec(labels, torch.int64, "labels"),
                                           _vec(label_rows, torch.int64, "label_rows", allow_none=True), float(grad_scale),
                                           ap, ld_agg, yp, ld_y, _vec(inv_norm, torch.float32, "inv_norm", allow_none=True),
                                           dp, ld_dz, _vec(row_loss, torch.float32, "row_loss"), gp, ld_dh, zp, ld_dpre,
                                           _p(writers, torch.int32, "writers", allow_none=True), _stream())
    _lib.check(rc, "sage_logit_layer_pool_fwd_bwd")


def count_row_writers(seg_ptr, src, m, writers, max_edges=None):
    """writers[r] = number of edges of the first m segments whose source is row r, + 1 for r < m
    (sage_count_row_writers; what logit_layer_fwd(dh=..., writers=...) wants)."""
    lib = _lib.load()
    m = int(m)
    if writers.dtype != torch.int32 or not writers.is_contiguous() or writers.numel() < m or seg_ptr.numel() < m + 1:
        raise RuntimeError("count_row_writers: bad buffers")
    rc = lib.sage_count_row_writers(_vec(seg_ptr, torch.int64, "seg_ptr"), _vec(src, torch.int64, "src"), m,
                                    int(src.numel() if max_edges is None else max_edges),
                                    _p(writers, torch.int32, "writers"), writers.numel(), _stream())
    _lib.check(rc, "sage_count_row_writers")
    return writers


def logit_layer_infer(h_rows, seg_ptr, src, m, reduce, w_l, w_r, y=None, pred=None, pre_activation=False):
    """The last SAGE layer on the m seed rows, forward only (reference models/graphsage.py:51-68 +
    the argmax of fig2a/runners/graphsage_runners.py:57-66): y [m, n] normalised logits and / or
    pred int64[m] = argmax over the classes (first maximum, as torch.argmax).  Returns (y, pred)."""
    lib = _lib.load()
    hp, ld_h = _mat(h_rows, "h_rows")
    wlp, ld_wl = _mat(w_l, "w_l")
    wrp, ld_wr = _mat(w_r, "w_r")
    n, h = w_l.shape
    m = int(m)
    if y is None and pred is None:
        y = torch.empty((m, n), dtype=torch.float32, device=h_rows.device)
    yp, ld_y = (None, 0) if y is None else _mat(y, "y")
    if h_rows.size(1) != h or tuple(w_r.shape) != (n, h) or m <= 0 or h_rows.size(0) < m \
            or seg_ptr.numel() < m + 1 or (y is not None and (y.size(0) < m or y.size(1) != n)) \
            or (pred is not None and pred.numel() < m):
        raise RuntimeError("logit_layer_infer: bad buffers")
    rc = lib.sage_logit_layer_infer(hp, ld_h, 1 if pre_activation else 0, _vec(seg_ptr, torch.int64, "seg_ptr"),
                                    _vec(src, torch.int64, "src"), m, h, REDUCE[reduce], wlp, ld_wl, wrp, ld_wr, n,
                                    yp, ld_y, _vec(pred, torch.int64, "pred", allow_none=True), _stream())
    _lib.check(rc, "sage_logit_layer_infer")
    return y, pred


def logit_layer_bwd(dz, seg_ptr, src, m, reduce, w_l, w_r, dh, row_loss=None, loss=None):
    """dh[src[e]] += (dz[t] @ w_l) (/ deg for mean), dh[t] += dz[t] @ w_r for the m seed rows, and
    loss = mean(row_loss[:m]) if both are given — one launch (sage_linear_dinput2 +
    sage_gather_reduce_bwd + the loss mean)."""
    lib = _lib.load()
    dp, ld_dz = _mat(dz, "dz")
    wlp, ld_wl = _mat(w_l, "w_l")
    wrp, ld_wr = _mat(w_r, "w_r")
    n, h = w_l.shape
    m = int(m)
    hp, ld_dh = _mat(dh, "dh")
    if tuple(w_r.shape) != (n, h) or dz.size(0) < m or dz.size(1) != n or dh.size(1) != h or dh.size(0) < m \
            or seg_ptr.numel() < m + 1 or (row_loss is None) != (loss is None) \
            or (row_loss is not None and row_loss.numel() < m):
        raise RuntimeError("logit_layer_bwd: bad buffers")
    rc = lib.sage_logit_layer_bwd(dp, ld_dz, _vec(seg_ptr, torch.int64, "seg_ptr"), _vec(src, torch.int64, "src"), m, h,
                                  REDUCE[reduce], wlp, ld_wl, wrp, ld_wr, n, hp, ld_dh,
                                  _vec(row_loss, torch.float32, "row_loss", allow_none=True),
                                  _p(loss, torch.float32, "loss", allow_none=True), _stream())
    _lib.check(rc, "sage_logit_layer_bwd")


def logit_layer_bwd_dw(dz, seg_ptr, src, m, reduce, w_l, w_r, dh, agg, dw_l, h_root, dw_r, row_loss=None, loss=None,
                       m_dev=None):
    """logit_layer_bwd plus the layer's weight gradients in the same launch:
    dw_l += dz[:m]^T @ agg[:m], dw_r += dz[:m]^T @ h_root[:m] (rows limited by m_dev if given)."""
    lib = _lib.load()
    dp, ld_dz = _mat(dz, "dz")
    wlp, ld_wl = _mat(w_l, "w_l")
    wrp, ld_wr = _mat(w_r, "w_r")
    n, h = w_l.shape
    m = int(m)
    hp, ld_dh = _mat(dh, "dh")
    ap, ld_agg = _mat(agg, "agg")
    rp, ld_hr = _mat(h_root, "h_root")
    glp, ld_gl = _mat(dw_l, "dw_l")
    grp, ld_gr = _mat(dw_r, "dw_r")
    if tuple(w_r.shape) != (n, h) or dz.size(0) < m or dz.size(1) != n or dh.size(1) != h or dh.size(0) < m \
            or seg_ptr.numel() < m + 1 or (row_loss is None) != (loss is None) \
            or (row_loss is not None and row_loss.numel() < m) \
            or agg.size(0) < m or agg.size(1) != h or h_root.size(0) < m or h_root.size(1) != h \
            or tuple(dw_l.shape) != (n, h) or tuple(dw_r.shape) != (n, h):
        raise RuntimeError("logit_layer_bwd_dw: bad buffers")
    rc = lib.sage_logit_layer_bwd_dw(dp, ld_dz, _vec(seg_ptr, torch.int64, "seg_ptr"), _vec(src, torch.int64, "src"), m,
                                     _count(m_dev, "m_dev"), h, REDUCE[reduce], wlp, ld_wl, wrp, ld_wr, n, hp, ld_dh,
                                     _vec(row_loss, torch.float32, "row_loss", allow_none=True),
                                     _p(loss, torch.float32, "loss", allow_none=True), ap, ld_agg, glp, ld_gl,
                                     rp, ld_hr, grp, ld_gr, _stream())
    _lib.check(rc, "sage_logit_layer_bwd_dw")


def adam_step(param, grad, exp_avg, exp_avg_sq, step_dev, lr, beta1=0.9, beta2=0.999, eps=1e-8,
              weight_decay=0.0, grad_scale=1.0, step_out=None, zero_grad=False, done_event=None, parts=None):
    """torch.optim.Adam step on flat fp32 buffers.  The step count is read from step_dev
    (int64[1]) and count+1 written to step_out (default: step_dev itself, which costs a second
    tiny kernel).  zero_grad=True clears grad after use.  done_event: a torch.cuda.Event that has
    been recorded at least once (its handle exists) — attached to the launch as its completion
    event instead of a record() behind it (sage_adam_step_signal).  parts = (ws, part_stride,
    nparts, region_off, region_len): the gradient of that region of the flat buffer is the sum of
    nparts partial buffers (linear_dweight2_parts; sage_adam_step_parts)."""
    lib = _lib.load()
    n = param.numel()
    for t, name in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        if t.numel() != n or not t.is_contiguous():
            raise RuntimeError(f"adam_step: {name} must be contiguous with {n} elements")
    args = (_p(param, torch.float32, "param"), _p(grad, torch.float32, "grad"),
            _p(exp_avg, torch.float32, "exp_avg"),
            _p(exp_avg_sq, torch.float32, "exp_avg_sq"), n, float(lr), float(beta1),
            float(beta2), float(eps), float(weight_decay), float(grad_scale),
            _count(step_dev, "step_dev"),
            _count(step_dev if step_out is None else step_out, "step_out"),
            1 if zero_grad else 0)
    handle = None
    if done_event is not None:
        handle = done_event.cuda_event
        if not handle:
            raise RuntimeError("adam_step: done_event has no handle yet (record it once first)")
    if parts is not None:
        ws, stride, nparts, off, length = parts
        if ws.dtype != torch.float32 or not ws.is_contiguous() or ws.numel() < (nparts - 1) * stride + length \
                or step_out is None:
            raise RuntimeError("adam_step: bad partial-sum buffers (and step_out must be a word of its own)")
        _lib.check(lib.sage_adam_step_parts(*args, _p(ws, torch.float32, "ws"), int(stride), int(nparts), int(off),
                                            int(length), ctypes.c_void_p(handle) if handle else None, _stream()),
                   "sage_adam_step_parts")
    elif handle is None:
        _lib.check(lib.sage_adam_step(*args, _stream()), "sage_adam_step")
    else:
        _lib.check(lib.sage_adam_step_signal(*args, ctypes.c_void_p(handle), _stream()), "sage_adam_step_signal")


# ------------------------------------------------------------------------------- a-7
def segment_slots(seg_ptr, src, num_targets, n_in, node_map=None):
    """Per target, the row ids of its first n_in inputs (padded with the first) + nonempty mask.
    slots is [n_in, T] (slot-major): slots[j] is the row map of input j over all targets."""
    lib = _lib.load()
    T = int(num_targets)
    if seg_ptr.numel() < T + 1:
        raise RuntimeError("segment_slots: seg_ptr shorter than num_targets+1")
    slots = torch.empty((int(n_in), T), dtype=torch.int64, device=seg_ptr.device)
    nonempty = torch.empty(T, dtype=torch.float32, device=seg_ptr.device)
    rc = lib.sage_segment_slots(_vec(seg_ptr, torch.int64, "seg_ptr"), _vec(src, torch.int64, "src"),
                                _vec(node_map, torch.int64, "node_map", allow_none=True), T, int(n_in),
                                _vec(slots.view(-1), torch.int64, "slots"), _vec(nonempty, torch.float32, "nonempty"),
                                _stream())
    _lib.check(rc, "sage_segment_slots")
    return slots, nonempty


def lstm_cell0_fwd(gates, nonempty, num_features, m=None, m_dev=None, bias_a=None, bias_b=None, out=None):
    """h = sigmoid(o) * tanh(sigmoid(i) * tanh(g)) from a zero state, zero rows where nonempty == 0
    (reference layers/sage.py:187).  bias_a / bias_b ([4F]): added to the gates on the way in."""
    lib = _lib.load()
    gp, ld_g = _mat(gates, "gates")
    m = gates.size(0) if m is None else int(m)
    if gates.size(1) != 4 * num_features or gates.size(0) < m or nonempty.numel() < m:
        raise RuntimeError("lstm_cell0_fwd: shape mismatch")
    for b in (bias_a, bias_b):
        if b is not None and b.numel() != 4 * num_features:
            raise RuntimeError("lstm_cell0_fwd: bias must have 4F elements")
    h = torch.empty((m, num_features), dtype=torch.float32, device=gates.device) if out is None else out
    if h.size(0) < m or h.size(1) != num_features:
        raise RuntimeError("lstm_cell0_fwd: bad output buffer")
    hp, ld_h = _mat(h, "h")
    rc = lib.sage_lstm_cell0_fwd(gp, ld_g, _vec(nonempty, torch.float32, "nonempty"), m, _count(m_dev, "m_dev"),
                                 int(num_features), _vec(bias_a, torch.float32, "bias_a", allow_none=True),
                                 _vec(bias_b, torch.float32, "bias_b", allow_none=True), hp, ld_h, _stream())
    _lib.check(rc, "sage_lstm_cell0_fwd")
    return h


def lstm_cell0_bwd(gates, nonempty, dh, num_features, m=None, m_dev=None, bias_a=None, bias_b=None, out=None):
    lib = _lib.load()
    gp, ld_g = _mat(gates, "gates")
    dp, ld_dh = _mat(dh, "dh")
    m = gates.size(0) if m is None else int(m)
    if gates.size(1) != 4 * num_features or gates.size(0) < m or dh.size(0) < m or dh.size(1) != num_features \
            or nonempty.numel() < m:
        raise RuntimeError("lstm_cell0_bwd: shape mismatch")
    for b in (bias_a, bias_b):
        if b is not None and b.numel() != 4 * num_features:
            raise RuntimeError("lstm_cell0_bwd: bias must have 4F elements")
    dgates = torch.empty_like(gates) if out is None else out
    if dgates.size(0) < m or dgates.size(1) != 4 * num_features:
        raise RuntimeError("lstm_cell0_bwd: bad output buffer")
    op, ld_dg = _mat(dgates, "dgates")
    rc = lib.sage_lstm_cell0_bwd(gp, ld_g, _vec(nonempty, torch.float32, "nonempty"), dp, ld_dh, m,
                                 _count(m_dev, "m_dev"), int(num_features),
                                 _vec(bias_a, torch.float32, "bias_a", allow_none=True),
                                 _vec(bias_b, torch.float32, "bias_b", allow_none=True), op, ld_dg, _stream())
    _lib.check(rc, "sage_lstm_cell0_bwd")
    return dgates


def lstm_gather_flat(x, seg_ptr, src, num_targets, n_in, x_flat, nonempty, node_map=None, num_targets_dev=None):
    """x_flat[t, j*F:(j+1)*F] = x[row(t, j)] (the cell's flattened input, reference
    layers/sage.py:164,:179) + nonempty[t]; buffers are the caller's."""
    lib = _lib.load()
    xp, ld_x = _mat(x, "x")
    fp, ld_f = _mat(x_flat, "x_flat")
    T, F = int(num_targets), x.size(1)
    if seg_ptr.numel() < T + 1 or x_flat.size(0) < T or x_flat.size(1) != n_in * F or nonempty.numel() < T:
        raise RuntimeError("lstm_gather_flat: shape mismatch")
    rc = lib.sage_lstm_gather_flat(xp, ld_x, _vec(node_map, torch.int64, "node_map", allow_none=True),
                                   _vec(seg_ptr, torch.int64, "seg_ptr"), _vec(src, torch.int64, "src"), T,
                                   _count(num_targets_dev, "num_targets_dev"), int(n_in), F, fp, ld_f,
                                   _vec(nonempty, torch.float32, "nonempty"), _stream())
    _lib.check(rc, "sage_lstm_gather_flat")
    return x_flat


def lstm_scatter_flat(d_flat, seg_ptr, src, num_targets, n_in, dx, num_targets_dev=None):
    """dx[row(t, j)] += d_flat[t, j*F:(j+1)*F] (backward of lstm_gather_flat without a node_map)."""
    lib = _lib.load()
    fp, ld_f = _mat(d_flat, "d_flat")
    xp, ld_dx = _mat(dx, "dx")
    T, F = int(num_targets), dx.size(1)
    if seg_ptr.numel() < T + 1 or d_flat.size(0) < T or d_flat.size(1) != n_in * F:
        raise RuntimeError("lstm_scatter_flat: shape mismatch")
    rc = lib.sage_lstm_scatter_flat(fp, ld_f, _vec(seg_ptr, torch.int64, "seg_ptr"), _vec(src, torch.int64, "src"), T,
                                    _count(num_targets_dev, "num_targets_dev"), int(n_in), F, xp, ld_dx, _stream())
    _lib.check(rc, "sage_lstm_scatter_flat")
    return dx


def lstm_pad_weight_sum(w, num_features, j0, j1, out):
    """out[r, f] = sum_{j0<=j<j1} w[r, j*F + f]: the weights every padded slot of a short segment
    shares (see sage_hip.h, sage_lstm_pad_weight_sum)."""
    lib = _lib.load()
    wp, ld_w = _mat(w, "w")
    op, ld_o = _mat(out, "out")
    F = int(num_features)
    if w.size(1) < j1 * F or out.size(0) != w.size(0) or out.size(1) != F or not 0 <= j0 <= j1:
        raise RuntimeError("lstm_pad_weight_sum: shape mismatch")
    rc = lib.sage_lstm_pad_weight_sum(wp, ld_w, w.size(0), F, int(j0), int(j1), op, ld_o, _stream())
    _lib.check(rc, "sage_lstm_pad_weight_sum")
    return out


def lstm_pad_grad_bcast(g, num_features, j0, j1, dw):
    """dw[r, j*F + f] += g[r, f] for j0<=j<j1; g comes back zeroed."""
    lib = _lib.load()
    gp, ld_g = _mat(g, "g")
    dp, ld_d = _mat(dw, "dw")
    F = int(num_features)
    if dw.size(1) < j1 * F or g.size(0) != dw.size(0) or g.size(1) != F or not 0 <= j0 <= j1:
        raise RuntimeError("lstm_pad_grad_bcast: shape mismatch")
    rc = lib.sage_lstm_pad_grad_bcast(gp, ld_g, dw.size(0), F, int(j0), int(j1), dp, ld_d, _stream())
    _lib.check(rc, "sage_lstm_pad_grad_bcast")
    return dw
