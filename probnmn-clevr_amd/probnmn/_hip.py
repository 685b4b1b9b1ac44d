"""ctypes binding of ``libprobnmn_hip.so`` (include/probnmn_hip.h).

The library is the product: if it is missing, or a call fails, this module raises -- there is no
CPU or eager-PyTorch fallback anywhere in the package.  ``torch`` is imported first on purpose:
the library links ``libamdhip64.so.7`` and must bind to the HIP runtime torch already loaded, so
that torch's stream handles and device pointers are valid inside it.
"""
import ctypes
import os
import time
from typing import Dict, Optional

import numpy as np
import torch  # noqa: F401  (must precede the dlopen below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PNMN_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libprobnmn_hip.so")

CHANNELS = 128
CONV_ACCUMULATE = 1
CONV_ATOMIC = 2
CONV_MASKBWD = 4
CONV_DATTN = 16


class HipLibraryError(RuntimeError):
    pass


_lib: Optional[ctypes.CDLL] = None

# name -> (restype, argtypes); kept in one table so tests can check every symbol the header
# declares is exported.
_P = ctypes.c_void_p
_I = ctypes.c_int
_D = ctypes.c_double
_F = ctypes.c_float
SIGNATURES: Dict[str, tuple] = {
    "pnmn_abi_version": (),
    "pnmn_conv_nhwc": (_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P),
    "pnmn_conv_nhwc_cus": (_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P),
    "pnmn_conv_wgrad_cus": (_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P),
    "pnmn_conv_wgrad": (_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P),
    "pnmn_transpose_weights": (_P, _I, _P),
    "pnmn_dot1_sigmoid_fwd": (_P, _I, _I, _P),
    "pnmn_dot1_sigmoid_bwd": (_P, _I, _I, _P),
    "pnmn_same_fwd": (_P, _I, _I, _P),
    "pnmn_same_bwd": (_P, _I, _I, _P),
    "pnmn_minmax_fwd": (_P, _I, _I, _I, _P),
    "pnmn_minmax_bwd": (_P, _I, _I, _I, _P),
    "pnmn_mask_bwd": (_P, _I, _I, _P),
    "pnmn_feat_grad_gather": (_P, _P, _I, _I, _I, _P),
    "pnmn_accumulate": (_P, _I, _P),
    "pnmn_nchw_to_nhwc": (_P, _P, _I, _I, _I, _P),
    "pnmn_nchw_to_nhwc_rows": (_P, _P, _P, _I, _I, _I, _P),
    "pnmn_nhwc_to_nchw": (_P, _P, _I, _I, _I, _P),
    "pnmn_copy_rows_h2d": (_P, _P, _P, _I, ctypes.c_int64, ctypes.c_int64, _P),
    "pnmn_gather_features": (_P, _P, _P, _I, ctypes.c_int64, _I, _I, _P),
    "pnmn_maxpool2_flatten_fwd": (_P, _P, _I, _I, _I, _I, _P),
    "pnmn_maxpool2_flatten_bwd": (_P, _P, _P, _I, _I, _I, _I, _P),
    "pnmn_answer_loss": (_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P),
    "pnmn_seq_nll_fwd": (_P, ctypes.c_int64, _P, ctypes.c_int64, _P, ctypes.c_int64, _I, _P, _P, _I, _I, _I, _F, _P),
    "pnmn_seq_nll_bwd": (_P, ctypes.c_int64, _P, ctypes.c_int64, _P, ctypes.c_int64, _I, _P, _P, _P, ctypes.c_int64,
                         _I, _I, _I, _F, _P),
    "pnmn_token_prep": (_P, ctypes.c_int64, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P),
    "pnmn_trim_predictions": (_P, _I, _I, _I, _P, _P),
    "pnmn_mask_last_fwd": (_P, _P, _P, _I, _I, _I, _P, _P, _P),
    "pnmn_mask_last_bwd": (_P, _P, _P, _P, _I, _I, _I, _P, _P),
    "pnmn_embedding_grad": (_P, _P, ctypes.c_int64, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P),
    "pnmn_embedding_grad_workspace_bytes": (_I, _I, _I),
    "pnmn_derive_params": (_P, _I, _I, _P),
    "pnmn_elbo_rows": (_P, _P, _P, _P, _P, _F, _F, _I, _P, _P, _P),
    "pnmn_joint_objective": (_P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P),
    "pnmn_clamp_adam": (_P, _I, _D, _D, _D, _D, _D, _D, _P),
    "pnmn_clamp_adam_blocks": (_P, _I, _D, _D, _D, _D, _D, _D, _I, _P),
    "pnmn_lstm_cell_fwd": (_P, _P, _P, _P, _P, _I, _I, _P),
    "pnmn_lstm_cell_bwd": (_P, _P, _P, _P, _P, _P, _P, _I, _I, _P),
    "pnmn_lstm_seq_fwd": (_P, _P, ctypes.c_int64, _P, _P, _P, _P, _I, _I, _I, _P, _P),
    "pnmn_lstm_seq_bwd": (_P, _P, _P, _P, _P, _I, _I, _I, _P, _P),
    "pnmn_token_table_fwd": (_P, _P, ctypes.c_int64, _P, _I, _I, _I, _P, _P),
    "pnmn_token_table_bwd": (_P, _P, _P, ctypes.c_int64, _I, _I, _I, _I, _P, _P, ctypes.c_int64, _P, _P, _P),
    "pnmn_token_rows": (_P, _I, _P, _I, ctypes.c_int64, _P),
    "pnmn_lstm_seq_workspace_bytes": (_I, _I),
    "pnmn_cluster_reserve_cus": (_I,),
    "pnmn_attn_lstm_fwd": (_P,) * 15 + (_I,) * 9 + (ctypes.c_uint64, ctypes.c_uint64, _P, ctypes.c_int64, _P),
    "pnmn_attn_lstm_bwd": (_P,) * 14 + (_I,) * 4 + (_P,),
    "pnmn_attn_lstm_multi_workspace_bytes": (_I, _I),
    "pnmn_attn_lstm_fwd_multi": (_P,) * 15 + (_I,) * 9 + (ctypes.c_uint64, ctypes.c_uint64, _P, ctypes.c_int64, _P, _P),
    "pnmn_attn_lstm_bwd_multi": (_P,) * 15 + (_I,) * 4 + (_P, _P),
    "pnmn_attn_lstm_pair_workspace_bytes": (_I, _I, _I),
    "pnmn_attn_lstm_fwd_multi_pair": (_P, _P, _I, _P, _P),
    "pnmn_attn_lstm_bwd_multi_pair": (_P, _P, _I, _P, _P),
    "pnmn_attn_lstm_group3_workspace_bytes": (_I, _I, _I, _I),
    "pnmn_attn_lstm_bwd_multi_group3": (_P, _P, _P, _I, _P, _P),
    "pnmn_attn_denc": (_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P),
    "pnmn_conv_nhwc_launches": (_I, _I, _I, _I, _I, _I),
    "pnmn_conv_force_split": (_I,),
    "pnmn_run_launches": (_P, _I, _P),
    "pnmn_conv2d_nhwc": (_P, _P),
    "pnmn_conv2d_weight_floats": (_I, _I, _I, _I),
    "pnmn_maxpool3x3s2_nhwc": (_P, _P, _I, _I, _I, _I, _P),
    "pnmn_launch_trace_begin": (),
    "pnmn_launch_trace_end": (_P, _I, _P),
    "pnmn_set_rows": (_P, _I, _P),
    "pnmn_trunk_planner_create": (_P, _P),
    "pnmn_trunk_planner_destroy": (_P,),
    "pnmn_trunk_plan_and_launch": (_P, _P, _P),
    "pnmn_trunk_last_forward": (_P, _P, _I),
    "pnmn_trunk_last_records_bytes": (_P, _P, ctypes.c_int64),
    "pnmn_plan_batch": (_P, _P, ctypes.c_int64, _P, _P, _I),
    "pnmn_compile_programs": (_P, _I, _I, _P, _I, _I, _P, _P, _P, _P),
    "pnmn_sample_tokens": (_P, _P, _P, _I, _I, _I, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, _I, _I, _I, _P),
    "pnmn_lstm_stack_workspace_bytes": (_P, _I, _I),
    "pnmn_lstm_stack_fwd": (_P, _I, _P, _P),
    "pnmn_lstm_stack_bwd": (_P, _I, _P, _P),
    "pnmn_gemm": (_P, _I, _P),
    "pnmn_gemm_cus": (_P, _I, _I, _P),
    "pnmn_gemm_workspace_bytes": (_I, _I, _I),
    "pnmn_gemm_split_k": (_I, _I, _I, _I),
    "pnmn_colsum": (_P, ctypes.c_int64, _I, _I, _P, _P, _I, _P, _P),
    "pnmn_colsum_workspace_bytes": (_I, _I),
}


ABI_VERSION = 12  # pnmn_abi_version() of the library these signatures describe (include/probnmn_hip.h)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                "libprobnmn_hip.so not found at %s -- build it with `python -c 'import "
                "__graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950); there is no "
                "fallback path" % LIB_PATH
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = ctypes.c_int64 if name.endswith("_bytes") else ctypes.c_int
            fn.argtypes = list(argtypes)
        if handle.pnmn_abi_version() != ABI_VERSION:  # (same symbol names, other argument lists: never call into it)
            raise HipLibraryError("%s has ABI version %d, this package binds version %d -- rebuild the library"
                                  % (LIB_PATH, handle.pnmn_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


_TRACE = None  # debugging aid (PNMN_TRACE_LAUNCHES=<seconds>): see _start_trace


def check(code: int, what: str) -> None:
    if code != 0:
        kind = "argument/shape error" if code < 0 else "hipError_t"
        raise HipLibraryError("%s failed: %s %d" % (what, kind, code))
    if _TRACE is not None:
        stream = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(stream)
        _TRACE.append((what, stream.cuda_stream, ev))
        del _TRACE[:-4000]


def mark(what: str) -> None:
    """A named point on the current stream's timeline (only when launch tracing is on)."""
    if _TRACE is not None:
        check(0, "mark: " + what)


def _start_trace(seconds: float) -> None:
    """Every library launch leaves an event behind; a watchdog thread reports, `seconds` after start, the
    first launch of each stream whose event has not completed -- i.e. the kernel a stalled GPU is stuck in
    (or waiting behind).  For diagnosing cross-stream stalls; costs one event per launch."""
    import sys
    import threading

    global _TRACE
    _TRACE = []

    def watch():
        time.sleep(seconds)
        per_stream = {}
        for what, stream, ev in list(_TRACE):
            st = per_stream.setdefault(stream, {"done": None, "stuck": None, "pending": 0})
            if ev.query():
                if st["stuck"] is None:
                    st["done"] = what
            else:
                st["pending"] += 1
                if st["stuck"] is None:
                    st["stuck"] = what
        for stream, st in per_stream.items():
            print("[pnmn trace] stream %#x: last completed launch = %s; first incomplete = %s (%d launches pending)"
                  % (stream, st["done"], st["stuck"], st["pending"]), file=sys.stderr, flush=True)

    threading.Thread(target=watch, daemon=True).start()


if os.environ.get("PNMN_TRACE_LAUNCHES"):
    _start_trace(float(os.environ["PNMN_TRACE_LAUNCHES"]))


# ---- item record layouts (must match include/probnmn_hip.h byte for byte) -------------------------
_u64 = np.uint64
_i32 = np.int32
CONV_ITEM = np.dtype(
    [("in", _u64), ("in2", _u64), ("mask", _u64), ("gate", _u64), ("weight", _u64), ("bias", _u64),
     ("out", _u64), ("dilation", _i32), ("flags", _i32), ("mb_feats", _u64), ("mb_attn", _u64),
     ("mb_dfeats", _u64), ("mb_dattn", _u64)]
)
WGRAD_ITEM = np.dtype(
    [("x", _u64), ("x2", _u64), ("xmask", _u64), ("dy", _u64), ("gate", _u64), ("dilation", _i32),
     ("reserved", _i32)]
)
WGRAD_JOB = np.dtype([("dw", _u64), ("dbias", _u64), ("item_begin", _i32), ("item_end", _i32)])
WTRANS_ITEM = np.dtype(
    [("src", _u64), ("dst", _u64), ("cout", _i32), ("cin", _i32), ("ntaps", _i32), ("reserved", _i32)]
)
DOT1_ITEM = np.dtype(
    [("in", _u64), ("w", _u64), ("b", _u64), ("out", _u64), ("dout", _u64), ("din", _u64),
     ("dw", _u64), ("db", _u64)]
)
SAME_ITEM = np.dtype(
    [("feats", _u64), ("attn", _u64), ("w", _u64), ("b", _u64), ("out", _u64), ("dout", _u64),
     ("dfeats", _u64), ("dattn", _u64), ("dw", _u64), ("db", _u64)]
)
MINMAX_ITEM = np.dtype(
    [("a", _u64), ("b", _u64), ("out", _u64), ("dout", _u64), ("da", _u64), ("db", _u64),
     ("a_channels", _i32), ("b_channels", _i32), ("is_max", _i32), ("reserved", _i32)]
)
MASKBWD_ITEM = np.dtype([("dx", _u64), ("feats", _u64), ("attn", _u64), ("dfeats", _u64), ("dattn", _u64)])
AXPY_ITEM = np.dtype([("src", _u64), ("dst", _u64), ("n", np.int64)])
DERIVE_JOB = np.dtype([("src", _u64), ("src2", _u64), ("dst", _u64), ("n", _i32), ("k", _i32), ("ld", _i32), ("kind", _i32)])
PLAN_IN = np.dtype([(n, _u64) for n in ("tables", "nprims", "tids", "examples", "base", "tokens", "w3", "b3", "wt3",
                                         "dotw", "dotb", "params", "grads", "wt", "act", "gact", "feat", "gfeat",
                                         "final_", "gfinal", "ones")]
                   + [(n, _i32) for n in ("n_templates", "pmax", "nv", "cmax", "hw", "channels", "wgrad_chunk",
                                          "wgrad_groups", "fuse_mask_bwd", "sole_writer", "sort_by_weight", "reserved")])
TRUNK_CONFIG = np.dtype([(n, _u64) for n in ("kinds", "w3", "b3", "wt3", "dotw", "dotb")]
                        + [(n, _i32) for n in ("n_kinds", "channels", "H", "W", "wgrad_chunk", "wgrad_groups", "fuse_mask_bwd",
                                               "sole_writer", "sort_by_weight", "reserved")])
TRUNK_IO = np.dtype([(n, _u64) for n in ("programs", "params", "grads", "wt", "act", "gact", "feat", "gfeat", "final_", "gfinal",
                                          "ones")]
                    + [("act_capacity", np.int64)]
                    + [(n, _u64) for n in ("fwd_tail", "bwd_head", "bwd_tail", "bwd", "valid")]
                    + [("arena_floats", np.int64)]
                    + [(n, _i32) for n in ("n_programs", "length", "n_fwd_tail", "n_bwd_head", "n_bwd_tail", "bwd_capacity",
                                           "need_backward", "launch", "n_bwd", "bwd_piece_cut", "n_prims", "n_fwd", "depth",
                                           "n_invalid", "n_feat_result", "conv_cus", "wgrad_cus", "n_conv", "n_proj", "reserved")]
                    + [("touched_tokens", _u64, (4,))])
DECODER_FWD_JOB = np.dtype([(n, _u64) for n in ("xe", "etable", "enc", "mask", "h0", "w_c", "w_hh", "w_p", "b_p", "hs", "cs", "act",
                                                  "ctx", "probs", "tokens", "in_tokens")]
                           + [("in_token_stride", np.int64), ("seed", _u64), ("row_offset", _u64)]
                           + [(n, _i32) for n in ("B", "T", "S", "V", "sample", "pad_index", "unk_index", "start_index")])
DECODER_BWD_JOB = np.dtype([(n, _u64) for n in ("dhs", "act", "cs", "hs", "probs", "enc", "mask", "h0", "w_c_t", "w_hh_t", "dgates",
                                                  "dctx", "dscore", "weights", "dh0")]
                           + [(n, _i32) for n in ("B", "T", "S", "reserved")])
GEMM_DESC = np.dtype([(n, _u64) for n in ("a", "b", "c", "bias")] + [(n, np.int64) for n in ("lda", "ldb", "ldc")]
                     + [(n, _i32) for n in ("M", "N", "K", "flags", "split_k", "shift_t")]
                     + [("shift_h0", _u64), ("ld_h0", np.int64), ("workspace", _u64), ("colsum", _u64), ("colsum2", _u64)])  # pnmn_gemm_desc
GEMM_MAX, GEMM_A_T, GEMM_B_T, GEMM_ACC = 8, 1, 2, 4
LSTM_STACK_JOB = np.dtype([("xp", _u64), ("tokens", _u64), ("token_stride", np.int64)] + [(n, _u64) for n in ("w_hh", "w_ih", "bias", "hs", "cs", "act", "dhs", "dgates")]
                          + [(n, _i32) for n in ("B", "T", "dep", "reserved")])  # pnmn_lstm_stack_job
LSTM_STACK_JOBS = 6
TOKEN_SEG = np.dtype([("src", _u64), ("index", _u64), ("row_stride", np.int64), ("rows", _i32), ("width", _i32)])  # pnmn_token_seg
EINVAL, ESHAPE, EAGAIN = -1, -2, -3  # PNMN_EINVAL / PNMN_ESHAPE / PNMN_EAGAIN
ADAM_ITEM = np.dtype([("param", _u64), ("grad", _u64), ("exp_avg", _u64), ("exp_avg_sq", _u64), ("n", np.int64),
                      ("bc1", np.float32), ("bc2_sqrt", np.float32)])

LAUNCH = np.dtype([("a", _u64), ("b", _u64), ("c", _u64), ("op", _i32), ("n", _i32), ("p", _i32, (8,))])
CONV2D_DESC = np.dtype([("x", _u64), ("w", _u64), ("scale", _u64), ("shift", _u64), ("residual", _u64), ("y", _u64),
                        ("N", _i32), ("H", _i32), ("W", _i32), ("Cin", _i32), ("Ho", _i32), ("Wo", _i32), ("Cout", _i32),
                        ("kh", _i32), ("kw", _i32), ("stride", _i32), ("pad", _i32), ("relu", _i32)])  # pnmn_conv2d_desc
LAUNCH_TIMING = np.dtype([("op", _i32), ("n", _i32), ("p", _i32, (8,)), ("n_items", _i32), ("ms", np.float32),
                          ("flops", np.float64), ("bytes", np.float64)])  # pnmn_launch_timing
(OP_CONV, OP_WGRAD, OP_TRANSPOSE_WEIGHTS, OP_DOT_FWD, OP_DOT_BWD, OP_SAME_FWD, OP_SAME_BWD, OP_MINMAX_FWD, OP_MINMAX_BWD,
 OP_MASK_BWD, OP_MAXPOOL_FWD, OP_MAXPOOL_BWD, OP_NCHW_TO_NHWC, OP_SET_ROWS, OP_ACCUMULATE, OP_ZERO, OP_FEAT_GATHER) = range(17)


class LaunchList:
    """A sequence of grouped launches for ``pnmn_run_launches`` (one binding call instead of one per launch).
    Rows are kept as tuples of eight 64-bit words -- the byte image of ``pnmn_launch`` (a, b, c, op | n << 32,
    p0 | p1 << 32, ...) -- and turned into one array when the list runs."""

    def __init__(self):
        self._rows = []

    def add(self, op: int, n: int, a: int, b: int = 0, c: int = 0, p=()) -> None:
        q = tuple(p) + (0,) * (8 - len(p))
        self._rows.append((a, b, c, op | (n << 32), q[0] | (q[1] << 32), q[2] | (q[3] << 32), q[4] | (q[5] << 32),
                           q[6] | (q[7] << 32)))

    def __len__(self) -> int:
        return len(self._rows)

    def run(self, stream: int, what: str) -> None:
        if self._rows:
            rec = np.array(self._rows, dtype=np.uint64)
            check(lib().pnmn_run_launches(rec.ctypes.data, len(self._rows), stream), what)
            self._rows = []


ITEM_SIZES = {
    "pnmn_launch": (LAUNCH, 64),
    "pnmn_launch_timing": (LAUNCH_TIMING, 64),
    "pnmn_conv2d_desc": (CONV2D_DESC, 96),
    "pnmn_conv_item": (CONV_ITEM, 96),
    "pnmn_wgrad_item": (WGRAD_ITEM, 48),
    "pnmn_wgrad_job": (WGRAD_JOB, 24),
    "pnmn_wtrans_item": (WTRANS_ITEM, 32),
    "pnmn_dot1_item": (DOT1_ITEM, 64),
    "pnmn_same_item": (SAME_ITEM, 80),
    "pnmn_minmax_item": (MINMAX_ITEM, 64),
    "pnmn_maskbwd_item": (MASKBWD_ITEM, 40),
    "pnmn_axpy_item": (AXPY_ITEM, 24),
    "pnmn_adam_item": (ADAM_ITEM, 48),
    "pnmn_derive_job": (DERIVE_JOB, 40),
    "pnmn_plan_in": (PLAN_IN, 216),
    "pnmn_decoder_fwd_job": (DECODER_FWD_JOB, 184),
    "pnmn_decoder_bwd_job": (DECODER_BWD_JOB, 136),
    "pnmn_trunk_config": (TRUNK_CONFIG, 88),
    "pnmn_trunk_io": (TRUNK_IO, 256),
    "pnmn_gemm_desc": (GEMM_DESC, 120),
    "pnmn_token_seg": (TOKEN_SEG, 32),
    "pnmn_lstm_stack_job": (LSTM_STACK_JOB, 104),
}


def stream_ptr(device: torch.device) -> int:
    """Raw handle of torch's current stream on ``device`` (every kernel call needs it: the raw query costs a
    fraction of building a ``torch.cuda.Stream`` object, ~6 us x 80 calls per step)."""
    idx = device.index
    if idx is None:
        idx = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx)


class _PinnedRing:
    """Reusable page-locked staging buffers for small host->device copies.

    ``Tensor.pin_memory()`` costs a hipHostMalloc (milliseconds) per call; a training step uploads
    its work lists every iteration, so staging memory is allocated once and recycled.  A slot is
    reused only after the copy that last read it has completed (event), which lets the host run
    several steps ahead of the GPU without overwriting bytes that are still to be copied.
    Two size classes: many fixed 16 KiB slots carved out of one allocation for the short records (optimiser
    items, derived-parameter jobs, index vectors: a copy is stream ordered, so waiting for a slot means
    waiting for everything queued before its last use -- with few slots the host could never run ahead), and a
    few growable slots for the per-step work lists."""

    SMALL, SMALL_SLOTS = 16 << 10, 64

    def __init__(self, slots: int = 6):
        self._bufs = [None] * slots
        self._events = [None] * slots
        self._next = 0
        self._small = None
        self._big = 16 << 20
        self._small_events = [None] * self.SMALL_SLOTS
        self._small_next = 0
        self.wait_seconds = 0.0  # time the host spent blocked on the GPU (diagnostic)

    def _wait(self, ev) -> None:
        if ev is not None and not ev.query():
            t0 = time.perf_counter()
            ev.synchronize()  # the host is far ahead of the GPU: wait for the slot
            self.wait_seconds += time.perf_counter() - t0

    def stage(self, raw, device: torch.device, total: int = -1) -> torch.Tensor:
        """``raw``: a uint8 array, or (with ``total`` = their summed size) a list of (offset, uint8 array) pieces that
        are copied straight into the pinned slot (no concatenated host copy first)."""
        pieces = None
        if total >= 0:
            pieces, n = raw, total
        else:
            n = raw.size
        if n <= self.SMALL:
            if self._small is None:
                self._small = torch.empty(self.SMALL * self.SMALL_SLOTS, dtype=torch.uint8).pin_memory()
            i = self._small_next
            self._small_next = (i + 1) % self.SMALL_SLOTS
            events, buf = self._small_events, self._small[i * self.SMALL:(i + 1) * self.SMALL]
        else:
            i = self._next
            self._next = (i + 1) % len(self._bufs)
            events, buf = self._events, self._bufs[i]
        self._wait(events[i])
        if buf is None or buf.numel() < n:
            # a hipHostMalloc costs milliseconds (19 ms seen for 0.5 MB): the growable slots share one size, 16 MB to
            # begin with -- several times the longest work list of any configuration measured (3 MB: 28x28 maps,
            # 40-token programs) -- that doubles when a list outgrows it; a slot is replaced when it is next used.
            # (Sized to the list at hand, every slot regrew whenever a step's sampled programs made a longer list than
            # that slot had seen: a 4 ms tail on one step in five at 128 questions.)
            while self._big < n:
                self._big *= 2
            buf = torch.empty(self._big, dtype=torch.uint8).pin_memory()
            self._bufs[i] = buf
        if pieces is None:
            buf.numpy()[:n] = raw
        else:
            view = buf.numpy()
            for off, piece in pieces:
                view[off:off + piece.size] = piece
        out = buf[:n].to(device, non_blocking=True)
        if events[i] is None:
            events[i] = torch.cuda.Event()
        events[i].record(torch.cuda.current_stream(device))
        return out


_rings: Dict[int, _PinnedRing] = {}


def _ring(device: torch.device) -> _PinnedRing:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    r = _rings.get(idx)
    if r is None:
        r = _rings[idx] = _PinnedRing()
    return r


def ring_wait_seconds() -> float:
    """Total time the host has been blocked waiting for staging slots (i.e. for the GPU)."""
    return sum(r.wait_seconds for r in _rings.values())


def to_device(records: np.ndarray, device: torch.device) -> torch.Tensor:
    """Copy a numpy (record) array into device memory asynchronously through the pinned ring.
    Returns a uint8 tensor owning the device copy; stream order protects it until the kernels that
    read it have run."""
    if device.type != "cuda":
        raise HipLibraryError("probnmn HIP kernels need a cuda (ROCm) device, got %s" % device)
    raw = np.ascontiguousarray(records).view(np.uint8).reshape(-1)
    return _ring(device).stage(raw, device)


def pieces_to_device(pieces, total: int, device: torch.device) -> torch.Tensor:
    """``to_device`` of the concatenation of (offset, uint8 array) pieces laid out in ``total`` bytes (gaps between
    pieces carry whatever the staging slot held)."""
    if device.type != "cuda":
        raise HipLibraryError("probnmn HIP kernels need a cuda (ROCm) device, got %s" % device)
    return _ring(device).stage(pieces, device, total)


def small_to_device(values, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    """A short Python list -> device tensor WITHOUT synchronising the stream (``torch.tensor(...,
    device=cuda)`` copies from pageable memory and waits for all queued work first)."""
    host = torch.tensor(values, dtype=dtype)
    if host.numel() == 0:
        return torch.empty(0, dtype=dtype, device=device)
    raw = host.view(torch.uint8).numpy() if dtype != torch.bool else host.numpy().view(np.uint8)
    return to_device(raw, device).view(dtype)
