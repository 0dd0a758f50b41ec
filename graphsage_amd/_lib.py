"""ctypes binding of libsage_hip.so (the C ABI declared in include/sage_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, a
RuntimeError is raised (the convention of the natives the reference binds through
torch.ops, e.g. samplers/uniform_loader.py:64).
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SAGE_HIP_LIB") or os.path.join(_HERE, "libsage_hip.so")   # override: A/B builds

# name -> (restype, argtypes); mirrors include/sage_hip.h declaration by declaration.
_P = c_void_p
SIGNATURES = {
    "sage_last_error": (c_char_p, []),
    "sage_abi_version": (c_int, []),
    "sage_probe_next_contraction": (c_int, [_P, _P, c_int]),
    "sage_event_record": (c_int, [_P, _P]),
    "sage_stream_wait_event": (c_int, [_P, _P]),
    "sage_csc_build_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sage_csc_build": (c_int, [_P, _P, c_int64, c_int64, _P, _P, _P, _P, c_size_t, _P]),
    "sage_sample_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sage_sample_fanout": (c_int, [_P, _P, c_int64, _P, c_int, ctypes.POINTER(c_int), c_int,
                                   c_uint64, c_uint64, _P, c_uint32, c_int64, c_int64,
                                   _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "sage_sample_fanout_ctrl": (c_int, [_P, _P, c_int64, _P, c_int, ctypes.POINTER(c_int), c_int,
                                        c_uint64, c_uint64, _P, c_uint32, _P, c_int64, c_int64,
                                        _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "sage_sample_fanout_static": (c_int, [_P, _P, c_int64, _P, c_int, ctypes.POINTER(c_int), c_int,
                                          c_uint64, c_uint64, _P, c_uint32, c_int64, c_int64,
                                          _P, _P, _P, _P, _P, _P, _P]),
    "sage_debug_exchange_wait": (c_int, [_P, c_int, _P, c_int, _P, _P]),
    "sage_ctrl_advance": (c_int, [_P, c_int64, c_int64, c_int64, c_int64, _P]),
    "sage_state_accumulate": (c_int, [_P, _P, _P]),
    "sage_random_nodes": (c_int, [c_int64, c_int64, c_uint64, c_uint64, _P, _P]),
    "sage_unsup_loss": (c_int, [_P, _P, _P, c_int64, c_int64, c_int, c_float, _P, _P, _P, c_int64, _P, _P, _P]),
    "sage_random_neighbor": (c_int, [_P, _P, c_int64, _P, c_int64, _P, c_uint64, c_uint64, _P, _P]),
    "sage_gather_rows": (c_int, [_P, c_int64, _P, c_int64, _P, c_int, _P, c_int64, _P]),
    "sage_gather_reduce_fwd": (c_int, [_P, c_int64, _P, _P, _P, c_int64, _P, c_int, c_int,
                                       _P, c_int64, _P, _P]),
    "sage_gather_reduce_fwd_timed": (c_int, [_P, c_int64, _P, _P, _P, c_int64, _P, c_int, c_int,
                                             _P, c_int64, _P, _P, _P, _P]),
    "sage_gather_reduce_bwd": (c_int, [_P, c_int64, _P, _P, _P, c_int64, _P, c_int, c_int,
                                       _P, c_int64, _P]),
    "sage_coo_to_csr_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sage_coo_to_csr": (c_int, [_P, _P, c_int64, c_int64, _P, _P, _P, _P, c_size_t, _P]),
    "sage_linear_concat_fwd": (c_int, [_P, c_int64, _P, c_int, _P, c_int64,
                                       _P, c_int64, _P, c_int, _P, c_int64,
                                       _P, c_int, c_int64, _P, c_int, _P, c_int64, _P]),
    "sage_linear_dinput": (c_int, [_P, c_int64, _P, c_int64, c_int64, _P, c_int, c_int,
                                   _P, c_int64, c_int, _P]),
    "sage_linear_dinput2": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64,
                                    c_int64, _P, c_int, c_int, _P]),
    "sage_linear_dweight": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int, c_int,
                                    _P, c_int64, _P, _P]),
    "sage_linear_dweight2": (c_int, [_P, c_int64, _P, c_int64, _P, c_int, _P, c_int64,
                                     _P, c_int64, _P, c_int, _P, c_int64, c_int64, _P, c_int, _P]),
    "sage_linear_dweight2_parts": (c_int, [_P, c_int64, _P, c_int64, _P, c_int, _P, c_int64,
                                     _P, c_int64, _P, c_int, _P, c_int64, c_int64, _P, c_int, c_int64, c_int, _P, _P]),
    "sage_segment_slots": (c_int, [_P, _P, _P, c_int64, c_int, _P, _P, _P]),
    "sage_lstm_cell0_fwd": (c_int, [_P, c_int64, _P, c_int64, _P, c_int, _P, _P, _P, c_int64, _P]),
    "sage_lstm_cell0_bwd": (c_int, [_P, c_int64, _P, _P, c_int64, c_int64, _P, c_int, _P, _P, _P, c_int64, _P]),
    "sage_lstm_gather_flat": (c_int, [_P, c_int64, _P, _P, _P, c_int64, _P, c_int, c_int, _P, c_int64, _P, _P]),
    "sage_lstm_pad_weight_sum": (c_int, [_P, c_int64, c_int64, c_int, c_int, c_int, _P, c_int64, _P]),
    "sage_lstm_pad_grad_bcast": (c_int, [_P, c_int64, c_int64, c_int, c_int, c_int, _P, c_int64, _P]),
    "sage_lstm_scatter_flat": (c_int, [_P, c_int64, _P, _P, c_int64, _P, c_int, c_int, _P, c_int64, _P]),
    "sage_l2norm_relu_fwd": (c_int, [_P, c_int64, c_int64, _P, c_int, _P, c_int64, _P, c_int64,
                                     _P, c_int, _P]),
    "sage_l2norm_relu_bwd": (c_int, [_P, c_int64, _P, c_int64, _P, c_int, c_int64, _P, c_int,
                                     _P, c_int64, c_int, _P]),
    "sage_l2norm_relu_bwd_z": (c_int, [_P, c_int64, _P, c_int64, c_int, c_int64, _P, c_int, _P, c_int64, c_int, _P]),
    "sage_l2norm_relu_bwd_z_dw": (c_int, [_P, c_int64, _P, c_int64, c_int, c_int64, _P, c_int, _P, c_int64, c_int,
                                          _P, c_int64, c_int64, _P, c_int, c_int, _P, c_int64, _P, c_int64,
                                          _P, c_int64, _P, c_int64, _P, _P, _P]),
    "sage_relu_bwd": (c_int, [_P, c_int64, _P, c_int64, c_int64, _P, c_int, _P, c_int64, c_int, _P]),
    "sage_pool_linear_fwd": (c_int, [_P, c_int64, _P, c_int, _P, c_int64, _P, c_int64, _P, c_int, _P, c_int64,
                                     _P, c_int64, _P]),
    "sage_pool_max_bwd": (c_int, [_P, c_int64, _P, c_int64, _P, _P, _P, c_int64, _P, c_int, _P, c_int64,
                                  c_int64, _P, _P, _P]),
    "sage_pool_mean_bwd": (c_int, [_P, c_int64, _P, c_int64, _P, _P, c_int64, _P, c_int, _P, c_int64, c_int64,
                                   _P, _P]),
    "sage_pool_reverse_edges_workspace_bytes": (c_size_t, [c_int64]),
    "sage_pool_reverse_edges": (c_int, [_P, _P, c_int64, _P, c_int64, _P, _P, _P, c_int64, _P, c_size_t, _P]),
    "sage_pool_max_fwd": (c_int, [_P, c_int64, _P, _P, c_int64, _P, c_int, c_int, _P, c_int64, _P, c_int64, _P]),
    "sage_pool_max_bwd_pull": (c_int, [_P, c_int64, _P, c_int64, _P, _P, c_int64, c_int64, _P, c_int, _P, c_int64, _P]),
    "sage_pool_mean_bwd_pull": (c_int, [_P, c_int64, _P, c_int64, _P, _P, _P, c_int64, c_int64, _P, c_int, _P, c_int64,
                                        _P]),
    "sage_softmax_xent": (c_int, [_P, c_int64, _P, _P, c_int64, c_int, c_float, _P, _P, c_int64,
                                  _P]),
    "sage_softmax_xent_rows": (c_int, [_P, c_int64, _P, _P, c_int64, c_int, c_float, _P, _P, _P, c_int64, _P]),
    "sage_logits_xent_fused": (c_int, [_P, c_int64, _P, _P, c_int64, c_int, c_float, _P, c_int64, _P,
                                       _P, c_int64, _P, _P, c_int, _P]),
    "sage_logits_bce_fused": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, c_int, c_float, _P, c_int64, _P, _P,
                                      c_int64, _P, _P, _P, c_int, _P]),
    "sage_logit_layer_fwd": (c_int, [_P, c_int64, _P, _P, c_int64, c_int, c_int, _P, c_int64, _P, c_int64, c_int,
                                     _P, _P, c_float, _P, c_int64, _P, c_int64, _P, _P, c_int64, _P, _P, c_int64, _P]),
    "sage_logit_layer_fwd_bwd": (c_int, [_P, c_int64, _P, _P, c_int64, c_int, c_int, _P, c_int64, _P, c_int64, c_int,
                                         _P, _P, c_float, _P, c_int64, _P, c_int64, _P, _P, c_int64, _P, _P, c_int64,
                                         _P, c_int64, _P, _P]),
    "sage_logit_layer_pool_fwd_bwd": (c_int, [_P, c_int64, _P, c_int64, _P, _P, c_int64, c_int, c_int, _P, c_int64, _P,
                                              c_int64, c_int, _P, _P, c_float, _P, c_int64, _P, c_int64, _P, _P, c_int64,
                                              _P, _P, c_int64, _P, c_int64, _P, _P]),
    "sage_count_row_writers": (c_int, [_P, _P, c_int64, c_int64, _P, c_int64, _P]),
    "sage_logit_layer_bwd": (c_int, [_P, c_int64, _P, _P, c_int64, c_int, c_int, _P, c_int64, _P, c_int64, c_int,
                                     _P, c_int64, _P, _P, _P]),
    "sage_logit_layer_bwd_dw": (c_int, [_P, c_int64, _P, _P, c_int64, _P, c_int, c_int, _P, c_int64, _P, c_int64, c_int,
                                        _P, c_int64, _P, _P, _P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P]),
    "sage_logit_layer_infer": (c_int, [_P, c_int64, c_int, _P, _P, c_int64, c_int, c_int, _P, c_int64, _P, c_int64,
                                       c_int, _P, c_int64, _P, _P]),
    "sage_adam_step": (c_int, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float,
                               c_float, c_float, _P, _P, c_int, _P]),
    "sage_adam_step_parts": (c_int, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float,
                                     c_float, c_float, _P, _P, c_int, _P, c_int64, c_int, c_int64, c_int64, _P, _P]),
    "sage_adam_step_signal": (c_int, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float,
                                      c_float, c_float, _P, _P, c_int, _P, _P]),
}

_lib = None
_tape = None      # list being recorded (see recording()), else None


class _Recorder:
    """Stands in for the CDLL while a tape is recorded: every entry point is called as usual
    and (function, arguments) is appended to the tape."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)

        def call(*args):
            _tape.append((fn, args, name))
            return fn(*args)
        return call


class recording:
    """Context manager: record the C-ABI calls made inside it.

    A static-shape caller (graphsage_amd/engine.py) issues the same calls with the same pointers
    every step; replaying the tape skips the per-call tensor validation and marshalling of
    graphsage_amd/ops.py — the host-side counterpart of a HIP graph, without the graph's
    per-node launch cost.  The arguments captured include the stream handle: a tape belongs to
    the stream it was recorded on.  `tape_py(fn)` adds a plain Python callable (a torch op that
    sits between two library calls) at the current position."""

    def __enter__(self):
        global _tape
        if _tape is not None:
            raise RuntimeError("nested recording")
        load()
        _tape = []
        return _tape

    def __exit__(self, *exc):
        global _tape
        _tape = None
        return False


def tape_py(fn):
    """Run fn() now and, while recording, remember it for replay."""
    if _tape is not None:
        _tape.append((fn, (), None))
    fn()


def replay(tape):
    for fn, args, name in tape:
        rc = fn(*args)
        if name is not None and rc != 0:
            check(rc, name)


def load():
    """Load libsage_hip.so once; raise RuntimeError if it is not built."""
    global _lib
    if _lib is not None:
        return _Recorder(_lib) if _tape is not None else _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C graphsage_amd/csrc` "
            "(or __graft_entry__.build()); graphsage_amd has no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError -> a symbol the header declares is missing
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return _Recorder(_lib) if _tape is not None else _lib


def last_error():
    return load().sage_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")
