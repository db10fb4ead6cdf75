"""ctypes binding of libghf_hip.so (the C ABI in include/ghf.h).

There is no CPU or eager-PyTorch fallback behind these functions: if the
library cannot be loaded the first call raises, loudly.  torch appears here
only to obtain device pointers and the current HIP stream.
"""

from __future__ import annotations

import ctypes as C
import os
import re
import threading
from typing import List, Optional, Sequence, Tuple

import torch

from . import _build

_lock = threading.Lock()
_lib: Optional[C.CDLL] = None

ABI_VERSION = 15
GHF_FLAG_NO_TAIL = 1
GHF_FLAG_RAW_SUM = 2
GHF_FLAG_ZERO_SRC = 4
GHF_FLAG_ZERO_DST = 8
GHF_FLAG_ADD_H = 16
SRC_MASK = (1 << 28) - 1        # sorted_src of block plans: node id below bit 28, run head above
WLAYOUT_NATURAL = 0
WLAYOUT_FRAG16 = 1
WLAYOUT_SPLIT2H = 3         # fp16 B fragments, 2 pieces per weight scaled by a power of two per relation (+ the scales)
SPLIT_LAYOUTS = (WLAYOUT_SPLIT2H,)   # layouts whose kernels gather pre-split rows (split_rows)

_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every function declared in include/ghf.h
SIGNATURES = {
    "ghf_abi_version": (_i32, []),
    "ghf_last_error": (C.c_char_p, []),
    "ghf_host_checksum64": (C.c_uint64, [_vp, _sz, C.c_uint64]),
    "ghf_host_word_ids": (C.c_longlong, [_vp, C.c_longlong, _vp, _vp, C.c_longlong, C.c_int]),
    "ghf_message_config": (_i32, [_i32, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "ghf_plan_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32, _i32]),
    "ghf_plan_max_chunks": (_i64, [_i64, _i64, _i32, _i32, _i32]),
    "ghf_plan_max_items": (_i64, [_i64, _i64, _i32, _i32, _i32, _i32]),
    "ghf_plan_build": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                              _vp, _vp]),
    "ghf_weightgen_fwd": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_vp), _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp,
                                 _vp, _vp, _vp, _vp, _vp, _vp]),
    "ghf_weightgen_fwd_batched": (_i32, [_i32, _vp, C.POINTER(_vp), C.POINTER(_vp), _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp,
                                         C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _vp]),
    "ghf_input_proj_fwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _i32, _vp]),
    "ghf_text_encode_fwd": (_i32, [_vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "ghf_message_layer_fwd": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _i32, _i32,
                                     _vp, _vp, _vp, _i32,
                                     _vp, _vp, _f32, _i64, _i64, _vp, _vp, _vp, _i32, _vp]),
    "ghf_message_side_output_supported": (_i32, [_i32, _i32, _i32]),
    "ghf_split_rows": (_i32, [_vp, _i64, _i32, _i64, _i64, _i32, _vp, _vp]),
    "ghf_split_rows_bytes": (_sz, [_i64, _i32, _i32]),
    "ghf_weights_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "ghf_group_workspace_bytes": (_sz, [_i64]),
    "ghf_group_edges": (_i32, [_vp, _i64, _i32, _vp, _sz, _vp, _vp, _vp]),
    "ghf_tail_bwd_workspace_floats": (_sz, [_i64, _i32]),
    "ghf_segment_axpy": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp]),
    "ghf_tail_bwd": (_i32, [_vp, _vp, _vp, _vp, _f32, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ghf_colsum_workspace_floats": (_sz, [_i64, _i32]),
    "ghf_colsum": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp]),
    "ghf_relu_mask": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "ghf_group_outer": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp]),
    "ghf_message_rs_supported": (_i32, [_i32]),
    "ghf_edge_transform_fwd": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "ghf_weights_rs_bytes": (_sz, [_i32, _i32]),
    "ghf_weights_pack_rs": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "ghf_edge_transform_h_fwd": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i32, _vp, _vp, _i64, _vp, _vp, _vp]),
    "ghf_run_rows_fwd": (_i32, [_vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp]),
    "ghf_segment_partial_fwd": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "ghf_segment_tail_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _i64, _i64, _i32, _vp, _vp, _i64, _i32, _vp]),
    "ghf_edge_outer_supported": (_i32, [_i32]),
    "ghf_edge_outer": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _vp]),
    "ghf_edge_outer_scaled": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _vp]),
    "ghf_scale_exp": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "ghf_add3": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "ghf_rowscale": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "ghf_dot": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "ghf_weightgen_acts": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ghf_weightgen_bwd_supported": (_i32, [_i32, _i32, _i32]),
    "ghf_weightgen_bwd_workspace_floats": (_sz, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "ghf_weightgen_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ghf_text_encode_bwd": (_i32, [_vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ghf_transpose_batched": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "ghf_weights_pack": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ghf_score_pairs_fwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp]),
    "ghf_rows_pack": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _vp, _vp]),
    "ghf_rows_unpack": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp]),
    "ghf_tail_fwd": (_i32, [_vp, _vp, _vp, _vp, _f32, _i64, _i64, _i32, _vp, _vp, _vp]),
    "ghf_set_range_flag": (_i32, [_vp]),
}


def header_symbols() -> List[str]:
    """Function names declared in include/ghf.h (for the export test)."""
    with open(os.path.join(_build.INCLUDE, "ghf.h")) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(ghf_[a-z_0-9]+)\s*\(", text)))


def lib_path() -> str:
    return _build.LIB_PATH


def load() -> C.CDLL:
    """Load libghf_hip.so (building it first if it is absent and hipcc is here)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = _build.LIB_PATH
        if not os.path.exists(path):
            try:
                _build.build()
            except Exception as exc:  # noqa: BLE001
                raise RuntimeError(
                    f"libghf_hip.so is missing at {path} and could not be built ({exc}). "
                    "This package has no CPU/PyTorch fallback: run __graft_entry__.build() first.") from exc
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.ghf_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libghf_hip.so ABI version {lib.ghf_abi_version()} != {ABI_VERSION}")
        if os.environ.get("GHF_TRACE_CALLS") == "1":          # diagnostics: name every C-ABI call on stderr and wait for it
            lib = _Traced(lib)
        _lib = lib
        return lib


class _Traced:
    """GHF_TRACE_CALLS=1: every entry point prints its name before it is enqueued and synchronises the device after — the last
    name on stderr is the call whose kernel faulted."""

    def __init__(self, lib) -> None:
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("ghf_") or name in ("ghf_last_error", "ghf_abi_version", "ghf_host_checksum64", "ghf_host_word_ids"):
            return fn

        def call(*args):
            import sys
            print(f"[ghf] {name}", file=sys.stderr, flush=True)
            rc = fn(*args)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            return rc
        return call


class GhfError(RuntimeError):
    pass


def _check(code: int, what: str) -> None:
    if code != 0:
        msg = load().ghf_last_error().decode("utf-8", "replace")
        if code == -1:
            raise ValueError(f"{what}: {msg}")
        raise GhfError(f"{what} failed (code {code}): {msg}")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    # the raw getter: torch.cuda.current_stream() without a device index walks through torch.cuda.is_available()
    # (environment lookups, ~40 us per launch on the GPU box — more than most of these kernels' launch cost)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _req(t: torch.Tensor, dtype: torch.dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a HIP device (got {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if t.device.index is not None and t.device.index != torch.cuda.current_device():
        # launches go to the CURRENT device's stream (_stream): a tensor elsewhere would be read from the wrong GPU's queue
        raise RuntimeError(f"{name} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                           "call torch.cuda.set_device(tensor.device) first (one process drives one GPU)")
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------
# thin wrappers (tensors in, tensors out; all work enqueued on the current stream)
# ---------------------------------------------------------------------------

# ---- range guard of the two-fp16-piece kernels (include/ghf.h: ghf_set_range_flag) -------------------------------
RANGE_ROWS, RANGE_WEIGHTS, RANGE_WEAK_W = 1, 2, 4
_range_flag: Optional[torch.Tensor] = None


def range_guard_enabled() -> bool:
    return os.environ.get("GHF_RANGE_GUARD", "1") != "0"


def range_flag(device) -> torch.Tensor:
    """The int32 device word the cutting kernels OR into (registered once per process)."""
    global _range_flag
    if _range_flag is None or _range_flag.device != torch.device(device):
        _range_flag = torch.zeros(1, dtype=torch.int32, device=device)
        _check(load().ghf_set_range_flag(_range_flag.data_ptr()), "ghf_set_range_flag")
    return _range_flag


class RangeFlagRead:
    """The guard word read EARLY: every bit a forward on the block kernels can raise is raised before its last layer is
    launched (rows are cut by the input projection and by the tails of the layers before the last; weights are packed
    before their layer starts), so the 4-byte copy is enqueued on a side stream behind an event recorded just before that
    launch, and the host waits for the copy while the GPU runs the last layer — it is back in time to enqueue the next
    forward, where a read at the very end left the GPU idle for the host's launch latency (C3: ~0.3 ms per forward)."""
    _host: Optional[torch.Tensor] = None
    _stream = None

    def __init__(self, flag: torch.Tensor) -> None:
        cls = RangeFlagRead
        if cls._host is None:
            cls._host = torch.zeros(1, dtype=torch.int32).pin_memory()
        if cls._stream is None or cls._stream.device != flag.device:
            cls._stream = torch.cuda.Stream(device=flag.device)
        self.flag, self.done = flag, None

    def arm(self) -> None:
        """Call on the forward's stream right before its last layer is enqueued."""
        cls = RangeFlagRead
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.flag.device))
        cls._stream.wait_event(ev)
        with torch.cuda.stream(cls._stream):
            cls._host.copy_(self.flag, non_blocking=True)
            self.done = torch.cuda.Event()
            self.done.record(cls._stream)

    def value(self) -> int:
        if self.done is None:                       # never armed (a path without an early point): a plain read
            return int(self.flag.item())
        self.done.synchronize()
        return int(RangeFlagRead._host[0])


def message_config(d: int, kernel: Optional[str] = None) -> Tuple[int, int, int, int]:
    """(block_nodes, weight layout, chunk_rows, split_chunks) of the message kernel for hidden size d; `kernel` names one
    as GHF_KERNEL would ("pp": the exact fp32-MFMA kernel, "generic", ...)."""
    bn, wl, cr, sc = _i32(0), _i32(0), _i32(0), _i32(0)
    lib = load()                                    # (before taking the lock: load() takes it too)
    with _lock:
        old = os.environ.get("GHF_KERNEL")
        try:
            if kernel is not None:
                os.environ["GHF_KERNEL"] = kernel
            _check(lib.ghf_message_config(int(d), C.byref(bn), C.byref(wl), C.byref(cr), C.byref(sc)), "ghf_message_config")
        finally:
            if kernel is not None:
                if old is None:
                    os.environ.pop("GHF_KERNEL", None)
                else:
                    os.environ["GHF_KERNEL"] = old
    return bn.value, wl.value, cr.value, sc.value


def exact_config(d: int) -> Tuple[int, int, int, int]:
    """The plan geometry of the exact (fp32 fma chain) kernel for hidden size d: the fp32-MFMA block kernel where one exists
    (d = 64, 128), else a CSR plan (generic kernel; relation-stationary layer on fp32 MFMAs for wide rows)."""
    cfg = message_config(d, "pp")
    return cfg if cfg[1] == WLAYOUT_FRAG16 else (1, WLAYOUT_NATURAL, 0, 0)


def plan_build(edge_index: torch.Tensor, rel_id: torch.Tensor, N: int, R: int, block_nodes: int, chunk_rows: int = 0,
               split_chunks: int = 0):
    """Returns a dict of the plan's device arrays (names as in include/ghf.h: ghf_plan_build) plus `status` [3]."""
    lib = load()
    ei = _req(edge_index, torch.int64, "edge_index")
    rel = _req(rel_id, torch.int64, "rel_id")
    E = ei.size(1)
    dev = ei.device
    nb = (N + block_nodes - 1) // block_nodes
    nseg = N if block_nodes == 1 else nb * R
    ws_bytes = lib.ghf_plan_workspace_bytes(N, E, R, block_nodes, chunk_rows)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)       # noqa: E731
    out = dict(sorted_key=i32(E), sorted_src=i32(E), seg_off=i32(nseg + 1), indeg=i32(N), status=i32(3),
               chunk_tab=None, blk_chunk_off=None, item_tab=None, blk_item_off=None)   # sorted_key: uint32 bit patterns
    if block_nodes > 1:
        out["chunk_tab"] = torch.zeros(2 * lib.ghf_plan_max_chunks(N, E, R, block_nodes, chunk_rows), dtype=torch.int32,
                                       device=dev)
        out["blk_chunk_off"] = i32(nb + 1)
        out["item_tab"] = torch.zeros(4 * lib.ghf_plan_max_items(N, E, R, block_nodes, chunk_rows, split_chunks),
                                      dtype=torch.int32, device=dev)
        out["blk_item_off"] = i32(nb + 1)
    _check(lib.ghf_plan_build(_ptr(ei), _ptr(rel), N, E, R, block_nodes, chunk_rows, split_chunks, _ptr(ws), ws_bytes,
                              _ptr(out["sorted_key"]), _ptr(out["sorted_src"]), _ptr(out["seg_off"]), _ptr(out["indeg"]),
                              _ptr(out["chunk_tab"]), _ptr(out["blk_chunk_off"]), _ptr(out["item_tab"]),
                              _ptr(out["blk_item_off"]), _ptr(out["status"]), _stream()), "ghf_plan_build")
    return out


def weightgen_fwd(text_emb: torch.Tensor, head_params: Sequence[torch.Tensor], log_scales: Sequence[torch.Tensor],
                  T: int, Hh: int, num_hidden: int, d_in: int, d_out: int, layout: int,
                  out: Optional[Tuple[torch.Tensor, Optional[torch.Tensor], torch.Tensor]] = None,
                  hidden_drop: Optional[torch.Tensor] = None, want_acts: bool = False):
    """head_params: flat list [head][layer][weight,bias]; log_scales: the three 1-element tensors (W_msg, W_self, bias),
    read in place; hidden_drop: scaled dropout masks [3, num_hidden, R, Hh] of the hidden activations (training);
    returns (W_msg or Wfrag, W_self or None, bias) — and, with want_acts, a fourth entry: the hidden activations
    [3, num_hidden, R, Hh] the same launch leaves for the backward (weightgen_acts' result; None without hidden layers)."""
    lib = load()
    x = _req(text_emb, torch.float32, "text_emb")
    R = x.size(0)
    dev = x.device
    keep = [_req(p, torch.float32, "weight-generator parameter") for p in head_params]
    arr = (_vp * len(keep))(*[p.data_ptr() for p in keep])
    if isinstance(log_scales, torch.Tensor):                  # a [3] tensor: three views of it
        log_scales = [log_scales[i:i + 1] for i in range(3)]
    ls_keep = [_req(t, torch.float32, "log_scale") for t in log_scales]
    if len(ls_keep) != 3:
        raise ValueError("weightgen_fwd: three log-scale tensors expected (W_msg, W_self, bias)")
    ls = (_vp * 3)(*[t.data_ptr() for t in ls_keep])
    hidden_ws = torch.empty(3 * 2 * R * max(Hh, T, 1), dtype=torch.float32, device=dev)
    if out is None:
        if layout != WLAYOUT_NATURAL:                # one opaque buffer holds [W_msg; W_self] in the kernel's order
            W_msg = torch.empty(lib.ghf_weights_bytes(R, d_in, d_out, layout) // 4, dtype=torch.float32, device=dev)
            W_self = None
        else:
            W_msg = torch.empty(R, d_in, d_out, dtype=torch.float32, device=dev)
            W_self = torch.empty(R, d_in, d_out, dtype=torch.float32, device=dev)
        bias = torch.empty(R, d_out, dtype=torch.float32, device=dev)
    else:
        W_msg, W_self, bias = out
    acts = torch.empty(3, num_hidden, R, Hh, dtype=torch.float32, device=dev) if want_acts and num_hidden > 0 else None
    _check(lib.ghf_weightgen_fwd(_ptr(x), arr, ls, R, T, Hh, num_hidden, d_in, d_out, layout,
                                 _ptr(hidden_ws), _ptr(W_msg), _ptr(W_self), _ptr(bias),
                                 _ptr(None if hidden_drop is None else _req(hidden_drop, torch.float32, "hidden_drop")), _ptr(acts),
                                 _stream()), "ghf_weightgen_fwd")
    return (W_msg, W_self, bias, acts) if want_acts else (W_msg, W_self, bias)


def weightgen_fwd_batched(text_emb: torch.Tensor, head_params: Sequence[Sequence[torch.Tensor]], log_scales: Sequence[Sequence[torch.Tensor]],
                          T: int, Hh: int, num_hidden: int, d_in: int, d_out: int, layout: int):
    """All L generators of a model in one launch sequence (include/ghf.h: ghf_weightgen_fwd_batched): head_params[g] /
    log_scales[g] as weightgen_fwd takes them; returns [(W_msg or Wfrag, W_self or None, bias)] per generator."""
    lib = load()
    x = _req(text_emb, torch.float32, "text_emb")
    R, dev, L = x.size(0), x.device, len(head_params)
    keep = [_req(p, torch.float32, "weight-generator parameter") for g in head_params for p in g]
    arr = (_vp * len(keep))(*[p.data_ptr() for p in keep])
    ls_keep = [_req(t, torch.float32, "log_scale") for g in log_scales for t in g]
    if len(ls_keep) != 3 * L:
        raise ValueError("weightgen_fwd_batched: three log-scale tensors per generator expected")
    ls = (_vp * len(ls_keep))(*[t.data_ptr() for t in ls_keep])
    hidden_ws = torch.empty(L * 3 * 2 * R * max(Hh, T, 1), dtype=torch.float32, device=dev)
    outs = []
    for _ in range(L):
        if layout != WLAYOUT_NATURAL:
            W_msg = torch.empty(lib.ghf_weights_bytes(R, d_in, d_out, layout) // 4, dtype=torch.float32, device=dev)
            W_self = None
        else:
            W_msg = torch.empty(R, d_in, d_out, dtype=torch.float32, device=dev)
            W_self = torch.empty(R, d_in, d_out, dtype=torch.float32, device=dev)
        outs.append((W_msg, W_self, torch.empty(R, d_out, dtype=torch.float32, device=dev)))
    wm = (_vp * L)(*[o[0].data_ptr() for o in outs])
    wsf = (_vp * L)(*[(None if o[1] is None else o[1].data_ptr()) for o in outs])
    bs = (_vp * L)(*[o[2].data_ptr() for o in outs])
    _check(lib.ghf_weightgen_fwd_batched(L, _ptr(x), arr, ls, R, T, Hh, num_hidden, d_in, d_out, layout, _ptr(hidden_ws),
                                         wm, wsf, bs, _stream()), "ghf_weightgen_fwd_batched")
    return outs


def text_encode_fwd(ids: torch.Tensor, lens: torch.Tensor, char_emb: torch.Tensor, W: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[U, Lmax] int32 ids + [U] lengths -> [U, T] text embeddings (reference TextEncoder, all strings at once)."""
    ids = _req(ids, torch.int32, "ids")
    lens = _req(lens, torch.int32, "lens")
    E, Wt, bt = (_req(t, torch.float32, n) for t, n in ((char_emb, "char_emb.weight"), (W, "proj.weight"), (b, "proj.bias")))
    U, Lmax = ids.shape
    out = torch.empty(U, Wt.size(0), dtype=torch.float32, device=ids.device)
    _check(load().ghf_text_encode_fwd(_ptr(ids), _ptr(lens), U, Lmax, _ptr(E), E.size(0), E.size(1), _ptr(Wt), _ptr(bt),
                                      Wt.size(0), _ptr(out), _stream()), "ghf_text_encode_fwd")
    return out


def input_proj_fwd(x: torch.Tensor, W_in: torch.Tensor, b_in: torch.Tensor, out: Optional[torch.Tensor] = None,
                   h_split: Optional[torch.Tensor] = None, split_layout: int = 0):
    lib = load()
    x = _req(x, torch.float32, "node_features")
    W = _req(W_in, torch.float32, "input_proj.weight")
    b = _req(b_in, torch.float32, "input_proj.bias")
    N, F = x.shape
    d = W.size(0)
    if W.size(1) != F:
        raise ValueError(f"node_features has {F} columns but input_proj expects {W.size(1)}")
    h0 = torch.empty(N, d, dtype=torch.float32, device=x.device) if out is None else out
    _check(lib.ghf_input_proj_fwd(_ptr(x), _ptr(W), _ptr(b), N, F, d, _ptr(h0), _ptr(h_split), split_layout, _stream()),
           "ghf_input_proj_fwd")
    return h0


def alloc_split(N: int, d: int, wlayout: int, device) -> torch.Tensor:
    """Uninitialised buffer for the split form of an [N, d] matrix (ghf_split_rows_bytes)."""
    nbytes = load().ghf_split_rows_bytes(N, d, wlayout)
    if nbytes == 0:
        raise ValueError(f"weight layout {wlayout} gathers h itself")
    return torch.empty((nbytes // 2,), dtype=torch.int16, device=device)


def split_rows(h: torch.Tensor, wlayout: int, out: Optional[torch.Tensor] = None, row0: int = 0,
               rows: Optional[int] = None) -> torch.Tensor:
    """Rows of h in the form the message kernel of `wlayout` gathers (include/ghf.h: ghf_split_rows): an opaque int16
    tensor — SPLIT2H: N*2*d fp16 bit patterns followed by N float scales."""
    lib = load()
    h = _req(h, torch.float32, "h")
    N, d = h.shape
    if out is None:
        out = alloc_split(N, d, wlayout, h.device)
    rows = N - row0 if rows is None else rows
    _check(lib.ghf_split_rows(_ptr(h), N, d, row0, rows, wlayout, _ptr(out), _stream()), "ghf_split_rows")
    return out


def message_layer_fwd(h: torch.Tensor, plan, W_msg: torch.Tensor, W_self: Optional[torch.Tensor],
                      bias: torch.Tensor, wlayout: int, ln_gamma: Optional[torch.Tensor],
                      ln_beta: Optional[torch.Tensor], ln_eps: float, h_out: torch.Tensor,
                      row0: int = 0, rows: Optional[int] = None, flags: int = 0,
                      h_split: Optional[torch.Tensor] = None, h_split_out: Optional[torch.Tensor] = None,
                      agg_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`plan` is a plan.GraphPlan: the device arrays, the host copy of the item offsets and the split-block scratch.
    SPLIT2H plans gather from `h_split` (split_rows(h, wlayout); made here when the caller has none) and can
    emit the split form of the rows they write into `h_split_out` for the next layer; `agg_out` (side_output_supported)
    also receives the aggregate before the tail."""
    lib = load()
    h = _req(h, torch.float32, "h")
    N, d = h.shape
    if rows is None:
        rows = N - row0
    item0, n_items, partial = plan.items_for(row0, rows, d)
    if wlayout in SPLIT_LAYOUTS and h_split is None:
        h_split = split_rows(h, wlayout)
    _check(lib.ghf_message_layer_fwd(_ptr(h), _ptr(h_split), N, d, _ptr(plan.sorted_key), _ptr(plan.sorted_src), _ptr(plan.seg_off),
                                     _ptr(plan.indeg), _ptr(plan.chunk_tab), _ptr(plan.blk_chunk_off), _ptr(plan.item_tab),
                                     _ptr(plan.blk_item_off), item0, n_items, _ptr(partial), plan.E, plan.R,
                                     plan.block_nodes, _ptr(W_msg), _ptr(W_self), _ptr(bias), wlayout, _ptr(ln_gamma),
                                     _ptr(ln_beta), float(ln_eps), row0, rows, _ptr(h_out), _ptr(h_split_out), _ptr(agg_out), flags,
                                     _stream()),
           "ghf_message_layer_fwd")
    return h_out


def side_output_supported(plan, d: int) -> bool:
    """Whether the plan's message kernel can also write the aggregate before the tail (message_layer_fwd's agg_out)."""
    return bool(load().ghf_message_side_output_supported(d, plan.block_nodes, plan.wlayout))


def score_pairs_fwd(a: torch.Tensor, b: torch.Tensor, ia: Optional[torch.Tensor] = None,
                    ib: Optional[torch.Tensor] = None) -> torch.Tensor:
    """scores[i] = a[ia[i]] . b[ib[i]] (indices optional): reference score_triple with its callers' row gathers fused."""
    a = _req(a, torch.float32, "a")
    b = _req(b, torch.float32, "b")
    if a.dim() != 2 or b.dim() != 2 or a.size(1) != b.size(1):
        raise ValueError(f"score_pairs: need two [rows, d] matrices of equal d, got {tuple(a.shape)} and {tuple(b.shape)}")
    ia = None if ia is None else _req(ia, torch.int64, "ia")
    ib = None if ib is None else _req(ib, torch.int64, "ib")
    n = ia.numel() if ia is not None else (ib.numel() if ib is not None else a.size(0))
    if (ia is not None and ib is not None and ia.numel() != ib.numel()) or (ia is None and ib is None and a.size(0) != b.size(0)):
        raise ValueError("score_pairs: the two sides name different numbers of pairs")
    out = torch.empty(n, dtype=torch.float32, device=a.device)
    _check(load().ghf_score_pairs_fwd(_ptr(a), _ptr(b), _ptr(ia), _ptr(ib), a.size(0), b.size(0), n, a.size(1), _ptr(out),
                                      _stream()), "ghf_score_pairs_fwd")
    return out


# ---- wide hidden sizes: relation-stationary layer (include/ghf.h, csrc/message_rs.hip) ---------------------------

def rs_supported(d: int) -> bool:
    return bool(load().ghf_message_rs_supported(d)) and os.environ.get("GHF_KERNEL") != "generic"


RS_MIN_RELATIONS = 160


def prefer_rs(d: int, R: int) -> bool:
    """Whether an inference plan for hidden size d and R relations should be a CSR plan for the relation-stationary layer:
    always where no destination-block kernel exists (d >= 256); at d = 128 from about 160 relations on — the block
    kernel re-streams a relation's weights per (block, relation) chunk, so its time grows with R while the
    relation-stationary layer's does not (tools/relation_sweep_rs.py, C3-sized graph, ms per layer, message_bx / relation-
    stationary: R = 64: 3.5 / 5.8, 96: 4.1 / 5.6, 128: 4.8 / 5.7, 192: 9.2 / 5.9, 256: 11.0 / 5.7)."""
    if not rs_supported(d) or os.environ.get("GHF_KERNEL") in ("bx", "pp"):
        return False
    return d >= 256 or R >= RS_MIN_RELATIONS or os.environ.get("GHF_KERNEL") in ("rs", "rs32")


def rs_exact(plan=None) -> bool:
    """The wide-row layer's pass 1 on fp32 MFMAs (exact fma chain) instead of two fp16 pieces: GHF_KERNEL=rs32, or a plan
    built for the exact kernels (GraphPlan.force_exact: the range guard's fallback) — carried by the plan, passed to
    edge_transform_fwd / segment_tail_fwd explicitly (no process-wide switch)."""
    return bool(plan is not None and getattr(plan, "force_exact", False)) or os.environ.get("GHF_KERNEL") == "rs32"


def edge_transform_fwd(h: torch.Tensor, rs, W_msg: torch.Tensor, W_self: torch.Tensor, bias: torch.Tensor, Y: torch.Tensor,
                       h_split: Optional[torch.Tensor] = None, exact: Optional[bool] = None) -> torch.Tensor:
    """Pass 1: per-edge results into Y [E, d] at the edges' destination-order positions (rs: plan.RsPlan; W_msg / W_self
    natural [R, d, d]).  Cuts the weights and — unless the caller has them (`h_split`, from the previous layer's pass 2) —
    the rows of h into their two fp16 pieces first (or transposes the weights, rs32).  Plans whose rows stand for runs of
    edges (rs.run_start: graphs with hubs) first sum every run's source rows (ghf_run_rows_fwd); the exact kernel has no
    such rows and runs the plan's per-edge twin."""
    lib = load()
    h = _req(h, torch.float32, "h")
    N, d = h.shape
    exact = rs_exact() if exact is None else (exact or rs_exact())
    if exact and rs.run_start is not None:
        rs = rs.per_edge()
        Y = rs.scratch(0, d, h.device)
    R = W_msg.size(0)
    Wm, Ws = _req(W_msg, torch.float32, "W_msg"), _req(W_self, torch.float32, "W_self")
    if exact:
        WmT, WsT = transpose_batched(Wm), transpose_batched(Ws)      # (named: they must outlive the launch's pointer taking)
        _check(lib.ghf_edge_transform_fwd(_ptr(h), N, d, _ptr(rs.src), _ptr(rs.dst), _ptr(rs.ypos), _ptr(rs.slice_tab),
                                          rs.slice_tab.size(0), _ptr(WmT), _ptr(WsT),
                                          _ptr(_req(bias, torch.float32, "bias")), _ptr(Y), _stream()), "ghf_edge_transform_fwd")
    else:
        hs = h_split if h_split is not None else split_rows(h, WLAYOUT_SPLIT2H)
        w2h = torch.empty(lib.ghf_weights_rs_bytes(R, d), dtype=torch.uint8, device=h.device)
        shift = torch.empty(R, dtype=torch.int32, device=h.device)
        _check(lib.ghf_weights_pack_rs(_ptr(Wm), _ptr(Ws), R, d, _ptr(w2h), _ptr(shift), _stream()), "ghf_weights_pack_rs")
        xs, nx = None, 0
        if rs.run_start is not None and rs.run_start.numel() > 1:
            nx = rs.run_start.numel() - 1
            xs = rs.run_scratch(nx, d)
            _check(lib.ghf_run_rows_fwd(_ptr(h), N, d, _ptr(rs.run_src), _ptr(rs.run_start), nx, _ptr(xs), _stream()), "ghf_run_rows_fwd")
        _check(lib.ghf_edge_transform_h_fwd(_ptr(hs), N, d, _ptr(rs.src), _ptr(rs.dst), _ptr(rs.ypos), _ptr(rs.slice_tab),
                                            rs.slice_tab.size(0), _ptr(w2h), R, _ptr(_req(bias, torch.float32, "bias")),
                                            _ptr(xs), nx, _ptr(rs.cnt), _ptr(Y), _stream()), "ghf_edge_transform_h_fwd")
    if rs.hub_of is not None:                       # hubs: their rows in chunks (fixed order); pass 2 adds the chunks' sums
        _check(load().ghf_segment_partial_fwd(_ptr(Y), _ptr(rs.hub_chunks), rs.hub_chunks.size(0), d, _ptr(rs.hub_scratch(d)),
                                              _stream()), "ghf_segment_partial_fwd")
    return Y


def segment_tail_fwd(Y: torch.Tensor, rs, h: Optional[torch.Tensor], ln_gamma, ln_beta, ln_eps: float, h_out: torch.Tensor,
                     row0: int = 0, rows: Optional[int] = None, flags: int = 0,
                     h_split_out: Optional[torch.Tensor] = None, exact: Optional[bool] = None) -> torch.Tensor:
    """Pass 2: destination sums of Y, mean and tail for rows [row0, row0+rows); `h_split_out` (alloc_split(N, d, SPLIT2H))
    also receives the rows in the form the next layer's pass 1 gathers."""
    N, d = h_out.shape
    rows = N - row0 if rows is None else rows
    exact = rs_exact() if exact is None else (exact or rs_exact())
    if exact and rs.run_start is not None:                        # (see edge_transform_fwd)
        rs = rs.per_edge()
        Y = rs.scratch(0, d, h_out.device)
    P = rs.hub_scratch(d) if rs.hub_of is not None else None      # filled by edge_transform_fwd
    _check(load().ghf_segment_tail_fwd(_ptr(Y), _ptr(rs.off), _ptr(rs.deg_of), _ptr(rs.hub_of), _ptr(rs.hub_tab), _ptr(P), _ptr(h), _ptr(ln_gamma),
                                       _ptr(ln_beta), float(ln_eps), row0, rows, d, _ptr(h_out), _ptr(h_split_out), N, flags,
                                       _stream()), "ghf_segment_tail_fwd")
    return h_out


# ---- backward pieces (include/ghf.h: "backward of the path") ----------------------------------------------------

def group_edges(rel_id: torch.Tensor, R: int):
    """(perm [E], goff [R+1]) int64: edge ids grouped stably by relation."""
    lib = load()
    rel = _req(rel_id, torch.int64, "rel_id")
    E = rel.numel()
    ws = torch.empty(lib.ghf_group_workspace_bytes(E), dtype=torch.uint8, device=rel.device)
    perm = torch.empty(E, dtype=torch.int64, device=rel.device)
    goff = torch.empty(R + 1, dtype=torch.int64, device=rel.device)
    _check(lib.ghf_group_edges(_ptr(rel), E, R, _ptr(ws), ws.numel(), _ptr(perm), _ptr(goff), _stream()), "ghf_group_edges")
    return perm, goff


def tail_bwd(grad_out: torch.Tensor, agg: torch.Tensor, h: torch.Tensor, gamma: torch.Tensor, eps: float, indeg: torch.Tensor,
             drop: Optional[torch.Tensor] = None, split_layout: Optional[int] = None):
    """(dpre, G, G_split, dgamma, dbeta) of include/ghf.h: ghf_tail_bwd (drop: the forward's scaled dropout mask, if any;
    split_layout: also return G cut as ``split_rows(G, split_layout)`` would, else G_split is None)."""
    lib = load()
    g = _req(grad_out, torch.float32, "grad_out")
    N, d = h.shape
    dpre, G = torch.empty_like(h), torch.empty_like(h)
    Gs = None if split_layout is None else alloc_split(N, d, split_layout, h.device)
    dgb = torch.empty(2, d, dtype=torch.float32, device=h.device)
    ws = torch.empty(max(int(lib.ghf_tail_bwd_workspace_floats(N, d)), 1), dtype=torch.float32, device=h.device)
    _check(lib.ghf_tail_bwd(_ptr(g), _ptr(_req(agg, torch.float32, "agg")), _ptr(_req(h, torch.float32, "h")),
                            _ptr(_req(gamma, torch.float32, "gamma")), float(eps), _ptr(indeg), N, d, _ptr(dpre), _ptr(G),
                            _ptr(Gs), _ptr(dgb), _ptr(ws), _ptr(None if drop is None else _req(drop, torch.float32, "drop")),
                            _stream()), "ghf_tail_bwd")
    return dpre, G, Gs, dgb[0], dgb[1]


def colsum(X: torch.Tensor, mask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[o] (+)= sum_v X[v][o] * (mask[v][o] > 0); accumulates when `out` is given."""
    lib = load()
    X = _req(X, torch.float32, "X")
    N, d = X.shape
    ws = torch.empty(lib.ghf_colsum_workspace_floats(N, d), dtype=torch.float32, device=X.device)
    acc = out is not None
    if out is None:
        out = torch.empty(d, dtype=torch.float32, device=X.device)
    _check(lib.ghf_colsum(_ptr(X), _ptr(None if mask is None else _req(mask, torch.float32, "mask")), N, d, _ptr(ws), _ptr(out),
                          1 if acc else 0, _stream()), "ghf_colsum")
    return out


def relu_mask(X: torch.Tensor, ref: torch.Tensor) -> torch.Tensor:
    X = _req(X, torch.float32, "X")
    out = torch.empty_like(X)
    _check(load().ghf_relu_mask(_ptr(X), _ptr(_req(ref, torch.float32, "ref")), X.numel(), _ptr(out), _stream()), "ghf_relu_mask")
    return out


def group_outer(A: Optional[torch.Tensor], ia: Optional[torch.Tensor], B: torch.Tensor, ib: Optional[torch.Tensor],
                goff: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C[g] (+)= sum_{e in goff[g]..goff[g+1]} A[ia[e]]^T (outer) B[ib[e]]; A None: column sums of the gathered B rows
    ([G, 1, db])."""
    lib = load()
    B = _req(B, torch.float32, "B")
    A = None if A is None else _req(A, torch.float32, "A")
    goff = _req(goff, torch.int64, "goff")
    da, db, ng = (0 if A is None else A.size(1)), B.size(1), goff.numel() - 1
    acc = out is not None
    if out is None:
        out = torch.empty(ng, max(da, 1), db, dtype=torch.float32, device=B.device)
    _check(lib.ghf_group_outer(_ptr(A), _ptr(ia), da, _ptr(B), _ptr(ib), db, goff.data_ptr(), goff.data_ptr() + 8, ng,
                               _ptr(out), 1 if acc else 0, _stream()), "ghf_group_outer")
    return out


def split_row_scales(split: torch.Tensor, N: int, d: int) -> torch.Tensor:
    """The N row scales (float32 view, no copy) behind the N rows of a SPLIT2H split form (split_rows / alloc_split)."""
    return split.view(torch.float32)[N * d:N * d + N]


def edge_outer(h: torch.Tensor, G: torch.Tensor, src: torch.Tensor, dst: torch.Tensor, slice_tab: torch.Tensor,
               slice_off: torch.Tensor, R: int, exact: bool = False, h_scales: Optional[torch.Tensor] = None,
               G_scales: Optional[torch.Tensor] = None, order: Optional[torch.Tensor] = None):
    """(dW [R, 2d, d] = dW_msg stacked on dW_self, db [R, d]) of include/ghf.h: ghf_edge_outer (exact: the fp32 chain at
    every d — a step that fell back to the exact kernels).  h_scales / G_scales: the row scales of the two tensors' split
    forms (split_row_scales), when the caller holds them: ghf_edge_outer_scaled, same bits without the pass over h and G.
    order [S] int32: the launch order of the slices (ghf.h; the same bits in any order)."""
    lib = load()
    h, G = _req(h, torch.float32, "h"), _req(G, torch.float32, "G")
    d, ns = h.size(1), slice_tab.size(0)
    D = min(d, 128)
    ws = torch.empty(ns * (2 * D * D + D) + 64, dtype=torch.float32, device=h.device)
    dW = torch.empty(R, 2 * d, d, dtype=torch.float32, device=h.device)
    db = torch.empty(R, d, dtype=torch.float32, device=h.device)
    tabs = (_ptr(_req(src, torch.int64, "src")), _ptr(_req(dst, torch.int64, "dst")), _ptr(_req(slice_tab, torch.int64, "slice_tab")),
            _ptr(_req(slice_off, torch.int64, "slice_off")), _ptr(_req(order, torch.int32, "order")) if order is not None else None)
    if order is not None and order.numel() != ns:
        raise ValueError("edge_outer: one launch position per slice expected")
    if not exact and h_scales is not None and G_scales is not None and h.size(0) > 0:
        hsc, gsc = _req(h_scales, torch.float32, "h_scales"), _req(G_scales, torch.float32, "G_scales")
        if hsc.numel() != h.size(0) or gsc.numel() != G.size(0):
            raise ValueError("edge_outer: one row scale per row of h and of G expected")
        _check(lib.ghf_edge_outer_scaled(_ptr(h), _ptr(G), _ptr(hsc), _ptr(gsc), *tabs, ns, R, d, h.size(0), _ptr(ws), _ptr(dW),
                                         _ptr(db), _stream()), "ghf_edge_outer_scaled")
        if os.environ.get("GHF_EO_GUARD_DEBUG"):             # (diagnostics: the range guard's counters of this call, ghf.h)
            print("edge_outer guard (far-down rows, nonzero rows) of h, G:", ws[ns * (2 * D * D + D) + 2:].view(torch.int32)[:4].tolist(), flush=True)
    else:
        _check(lib.ghf_edge_outer(_ptr(h), _ptr(G), *tabs, ns, R, d, 0 if exact else h.size(0), _ptr(ws), _ptr(dW), _ptr(db),
                                  _stream()), "ghf_edge_outer")
    return dW, db


_RANGES: dict = {}


def _row_ranges(K: int, step: int, device) -> torch.Tensor:
    """[0, step, 2 step, ..., K] on the device (cached: a handful of distinct sizes per model)."""
    key = (K, step, str(device))
    r = _RANGES.get(key)
    if r is None:
        n = (K + step - 1) // step
        r = torch.tensor([min(i * step, K) for i in range(n + 1)], dtype=torch.int64).to(device)
        if len(_RANGES) >= 64:
            _RANGES.pop(next(iter(_RANGES)))
        _RANGES[key] = r
    return r


def matmul_tn(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """A^T B for row-major A [K, M], B [K, N] -> [M, N] in exact fp32: ghf_group_outer over slices of the K rows (so that
    a tall contraction fills the chip), the slices' partial products summed in order by ghf_colsum."""
    K, M = A.shape
    N = B.size(1)
    tiles = ((M + 15) // 16) * ((N + 127) // 128)
    step = K if K <= 512 else max(256, ((K * tiles + 2047) // 2048 + 3) & ~3)      # aim at ~2000 workgroups
    goff = _row_ranges(K, step, A.device)
    part = group_outer(A, None, B, None, goff)
    if part.size(0) == 1:
        return part[0]
    return colsum(part.view(part.size(0), M * N)).view(M, N)


def matmul_nn(X: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """X W for row-major X [M, K], W [K, N] (the gradient of a Linear with respect to its input)."""
    return matmul_tn(transpose_batched(X.unsqueeze(0))[0], W)


def scale_exp(X: torch.Tensor, log_scale: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    X = _req(X, torch.float32, "X")
    out = torch.empty_like(X) if out is None else out
    _check(load().ghf_scale_exp(_ptr(X), X.numel(), _ptr(_req(log_scale, torch.float32, "log_scale")), _ptr(out), _stream()),
           "ghf_scale_exp")
    return out


def add3(a: torch.Tensor, b: torch.Tensor, c: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    a, b = _req(a, torch.float32, "a"), _req(b, torch.float32, "b")
    c = None if c is None else _req(c, torch.float32, "c")
    if a.numel() != b.numel() or (c is not None and c.numel() != a.numel()):
        raise ValueError("add3: operands differ in size")
    out = torch.empty_like(a) if out is None else out
    _check(load().ghf_add3(_ptr(a), _ptr(b), _ptr(c), a.numel(), _ptr(out), _stream()), "ghf_add3")
    return out


def rowscale(X: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    X, g = _req(X, torch.float32, "X"), _req(g, torch.float32, "g")
    n, d = X.shape
    if g.numel() != n:
        raise ValueError("rowscale: one factor per row expected")
    out = torch.empty_like(X)
    _check(load().ghf_rowscale(_ptr(X), _ptr(g), n, d, _ptr(out), _stream()), "ghf_rowscale")
    return out


def segment_axpy(w: torch.Tensor, iw: torch.Tensor, X: torch.Tensor, ix: torch.Tensor, off: torch.Tensor) -> torch.Tensor:
    """out[v] = sum_{e in off[v]..off[v+1]} w[iw[e]] * X[ix[e]] (include/ghf.h: ghf_segment_axpy)."""
    w, X = _req(w, torch.float32, "w"), _req(X, torch.float32, "X")
    iw, ix, off = _req(iw, torch.int64, "iw"), _req(ix, torch.int64, "ix"), _req(off, torch.int64, "off")
    if iw.numel() != ix.numel():
        raise ValueError("segment_axpy: one weight index per row index expected")
    nseg, (nx, d) = off.numel() - 1, X.shape
    out = torch.empty(nseg, d, dtype=torch.float32, device=X.device)
    _check(load().ghf_segment_axpy(_ptr(w), _ptr(iw), _ptr(X), _ptr(ix), _ptr(off), nseg, nx, d, _ptr(out), _stream()), "ghf_segment_axpy")
    return out


def dot(X: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
    """[1] tensor: sum of the elementwise product (fixed summation order)."""
    X, Y = _req(X, torch.float32, "X"), _req(Y, torch.float32, "Y")
    n = X.numel()
    if Y.numel() != n:
        raise ValueError("dot: operands differ in size")
    ws = torch.empty((n + 8191) // 8192, dtype=torch.float32, device=X.device)
    out = torch.empty(1, dtype=torch.float32, device=X.device)
    _check(load().ghf_dot(_ptr(X), _ptr(Y), n, _ptr(ws), _ptr(out), _stream()), "ghf_dot")
    return out


def weightgen_acts(text_emb: torch.Tensor, head_params: Sequence[torch.Tensor], T: int, Hh: int, num_hidden: int,
                   hidden_drop: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[3, num_hidden, R, Hh]: the hidden activations of the three generator heads (post-ReLU, and post-dropout when the
    forward's masks are given)."""
    x = _req(text_emb, torch.float32, "text_emb")
    keep = [_req(p, torch.float32, "weight-generator parameter") for p in head_params]
    arr = (_vp * len(keep))(*[p.data_ptr() for p in keep])
    acts = torch.empty(3, num_hidden, x.size(0), Hh, dtype=torch.float32, device=x.device)
    _check(load().ghf_weightgen_acts(_ptr(x), arr, x.size(0), T, Hh, num_hidden, _ptr(acts),
                                     _ptr(None if hidden_drop is None else _req(hidden_drop, torch.float32, "hidden_drop")), _stream()),
           "ghf_weightgen_acts")
    return acts


def weightgen_bwd_supported(T: int, Hh: int, num_hidden: int) -> bool:
    return bool(load().ghf_weightgen_bwd_supported(T, Hh, num_hidden))


def weightgen_bwd(text_emb: torch.Tensor, head_params: Sequence[torch.Tensor], acts: Optional[torch.Tensor],
                  outs: Sequence[torch.Tensor], grads: Sequence[torch.Tensor], log_scales: torch.Tensor, T: int, Hh: int,
                  num_hidden: int, d_in: int, d_out: int, log_keep: Optional[torch.Tensor] = None, want_dx: bool = True):
    """(dparams [one tensor per parameter], dls [3], d text_emb or None): include/ghf.h, ghf_weightgen_bwd — the three heads'
    backward in three launches.  log_scales: [3] float32 on the device."""
    lib = load()
    x = _req(text_emb, torch.float32, "text_emb")
    R, dev = x.size(0), x.device
    keep = [_req(p, torch.float32, "weight-generator parameter") for p in head_params]
    o = [_req(t, torch.float32, "generator output") for t in outs]
    g = [_req(t, torch.float32, "generator output gradient") for t in grads]
    ls = _req(log_scales, torch.float32, "log_scales")
    dparams = [torch.empty_like(p) for p in keep]
    dls = torch.empty(3, dtype=torch.float32, device=dev)
    dx = torch.empty(R, T, dtype=torch.float32, device=dev) if want_dx else None
    ws = torch.empty(lib.ghf_weightgen_bwd_workspace_floats(R, T, Hh, num_hidden, d_in, d_out), dtype=torch.float32, device=dev)
    arr = lambda ts: (_vp * len(ts))(*[t.data_ptr() for t in ts])       # noqa: E731
    ls_ptrs = (_vp * 3)(*[ls.data_ptr() + 4 * k for k in range(3)])
    dls_ptrs = (_vp * 3)(*[dls.data_ptr() + 4 * k for k in range(3)])
    _check(lib.ghf_weightgen_bwd(_ptr(x), arr(keep), _ptr(None if acts is None else _req(acts, torch.float32, "acts")), arr(o), arr(g),
                                 ls_ptrs, R, T, Hh, num_hidden, d_in, d_out,
                                 _ptr(None if log_keep is None else _req(log_keep, torch.float32, "log_keep")), arr(dparams), dls_ptrs,
                                 _ptr(dx), _ptr(ws), _stream()), "ghf_weightgen_bwd")
    return dparams, dls, dx


def text_encode_bwd(ids: torch.Tensor, lens: torch.Tensor, char_emb: torch.Tensor, W: torch.Tensor, te: torch.Tensor,
                    dte: torch.Tensor):
    """(d char_emb [V, C], d W [T, C], d b [T]) of ghf_text_encode_fwd."""
    E, Wt = _req(char_emb, torch.float32, "char_emb.weight"), _req(W, torch.float32, "proj.weight")
    te, dte = _req(te, torch.float32, "te"), _req(dte, torch.float32, "dte")
    U, Lmax = ids.shape
    V, Cd = E.shape
    T = Wt.size(0)
    ws = torch.empty(2 * U * Cd + U * T, dtype=torch.float32, device=E.device)
    dE, dW, db = torch.empty_like(E), torch.empty_like(Wt), torch.empty(T, dtype=torch.float32, device=E.device)
    _check(load().ghf_text_encode_bwd(_ptr(_req(ids, torch.int32, "ids")), _ptr(_req(lens, torch.int32, "lens")), U, Lmax,
                                      _ptr(E), V, Cd, _ptr(Wt), T, _ptr(te), _ptr(dte), _ptr(ws), _ptr(dE), _ptr(dW), _ptr(db),
                                      _stream()), "ghf_text_encode_bwd")
    return dE, dW, db


def transpose_batched(x: torch.Tensor) -> torch.Tensor:
    """[B, r, c] -> [B, c, r] (a copy)."""
    x = _req(x, torch.float32, "x")
    Bn, r, c = x.shape
    out = torch.empty(Bn, c, r, dtype=torch.float32, device=x.device)
    _check(load().ghf_transpose_batched(_ptr(x), Bn, r, c, _ptr(out), _stream()), "ghf_transpose_batched")
    return out


def weights_pack(top: Optional[torch.Tensor], bottom: Optional[torch.Tensor], transpose: bool, R: int, d: int, wlayout: int):
    """[top[r]; bottom[r]] (natural [R, d, d] each, None = zeros, optionally transposed) in the kernel's weight layout."""
    lib = load()
    dev = (top if top is not None else bottom).device
    out = torch.empty(lib.ghf_weights_bytes(R, d, d, wlayout) // 4, dtype=torch.float32, device=dev)
    _check(lib.ghf_weights_pack(_ptr(None if top is None else _req(top, torch.float32, "top")),
                                _ptr(None if bottom is None else _req(bottom, torch.float32, "bottom")), 1 if transpose else 0,
                                R, d, wlayout, _ptr(out), _stream()), "ghf_weights_pack")
    return out


def tail_fwd(agg: torch.Tensor, h: torch.Tensor, ln_gamma: torch.Tensor, ln_beta: torch.Tensor, ln_eps: float,
             h_out: torch.Tensor, row0: int = 0, rows: Optional[int] = None, drop: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = load()
    N, d = h.shape
    if rows is None:
        rows = N - row0
    _check(lib.ghf_tail_fwd(_ptr(agg), _ptr(h), _ptr(ln_gamma), _ptr(ln_beta), float(ln_eps), row0, rows, d,
                            _ptr(h_out), _ptr(None if drop is None else _req(drop, torch.float32, "drop")), _stream()), "ghf_tail_fwd")
    return h_out


def rows_pack(tables, idx: torch.Tensor) -> torch.Tensor:
    """The rows `idx` (int64, device) of one or two row-indexed byte tables ([N, row_bytes] uint8 views of one allocation each)
    as one contiguous message [n, row_bytes (+ extra_bytes)] uint8 — ghf_rows_pack."""
    rows = tables[0]
    extra = tables[1] if len(tables) > 1 else None
    rb, xb = rows.size(1) * rows.element_size(), (extra.size(1) * extra.element_size() if extra is not None else 0)
    out = torch.empty(idx.numel(), rb + xb, dtype=torch.uint8, device=rows.device)
    _check(load().ghf_rows_pack(_ptr(rows), rb, _ptr(extra), xb, _ptr(idx), idx.numel(), rows.size(0), _ptr(out), _stream()), "ghf_rows_pack")
    return out


def rows_unpack(tables, idx: torch.Tensor, packed: torch.Tensor) -> None:
    """ghf_rows_unpack: a received message back to the rows' places in the same tables."""
    rows = tables[0]
    extra = tables[1] if len(tables) > 1 else None
    rb, xb = rows.size(1) * rows.element_size(), (extra.size(1) * extra.element_size() if extra is not None else 0)
    if packed.numel() != idx.numel() * (rb + xb):
        raise ValueError(f"rows_unpack: message of {packed.numel()} bytes for {idx.numel()} rows of {rb + xb}")
    _check(load().ghf_rows_unpack(_ptr(packed), _ptr(idx), idx.numel(), rows.size(0), _ptr(rows), rb, _ptr(extra), xb, _stream()), "ghf_rows_unpack")
