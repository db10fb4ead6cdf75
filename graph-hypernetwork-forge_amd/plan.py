"""Graph plan: relation ids + the sorted edge arrays the HIP kernels consume, and its cache.

Replaces the per-forward host work of the reference (models/hypergnn.py:264-268:
dict.fromkeys dedupe, id list, torch.tensor) with a plan that is built once per
(edge_index, edge_texts) pair and reused: the `List[str]` boundary costs ~1.5 s
at 10 M edges, far more than the whole device forward.
"""

from __future__ import annotations

import ctypes
import operator
import os
import sys
from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native


MAX_FAST_OBJECTS = 1 << 16     # relation_ids' fast path: lists that reference at most this many distinct string OBJECTS


def relation_ids(edge_texts: Sequence[str], want_objects: bool = False):
    """Unique relation strings in first-appearance order and the per-edge id (int64).

    Same mapping as the reference (models/hypergnn.py:264-268); the order of the
    unique list only permutes the generated weights, never the result.  A knowledge graph's list usually references a few
    string objects millions of times: those lists are mapped by object identity first (ghf_host_word_ids over the list's
    pointer array, on a few host threads: a pass at memory speed instead of ten million dict lookups — 0.8 s -> 0.005 s at
    10 M edges), and only the distinct objects go through the reference's value-keyed dict.
    want_objects: a third result — the distinct string OBJECTS of that fast path (None when it was not taken): while a
    caller holds them, no other object can take one of their addresses (plan cache: PlanCache.put)."""
    n = len(edge_texts)
    base = _item_array_address(edge_texts) if n >= 4096 else None
    if base is not None:
        try:
            lib = _native.load()
        except Exception:                                    # (no library: the plain path below needs none)
            lib = None
        if lib is not None:
            ids = np.empty(n, dtype=np.int64)
            uniq = (ctypes.c_void_p * MAX_FAST_OBJECTS)()
            k = lib.ghf_host_word_ids(base, n, ids.ctypes.data, ctypes.addressof(uniq), MAX_FAST_OBJECTS, _VERIFY_THREADS)
            if k >= 0:
                objs = [ctypes.cast(uniq[i], ctypes.py_object).value for i in range(k)]      # (alive: the list holds them)
                unique = list(dict.fromkeys(objs))
                if len(unique) != k:                         # distinct objects with equal strings share an id
                    lut = {t: i for i, t in enumerate(unique)}
                    remap = np.fromiter((lut[o] for o in objs), dtype=np.int64, count=k)
                    ids = remap[ids]
                return (unique, ids, objs) if want_objects else (unique, ids)
    unique = list(dict.fromkeys(edge_texts))
    lut = {t: i for i, t in enumerate(unique)}
    ids = np.fromiter(map(lut.__getitem__, edge_texts), dtype=np.int64, count=len(edge_texts))
    return (unique, ids, None) if want_objects else (unique, ids)


@dataclass
class RsPlan:
    """What the relation-stationary layer (csrc/message_rs.hip, wide hidden sizes) reads besides the CSR plan.  A ROW of its
    two passes is an edge — or, in plans for graphs with hubs (run_start is not None), a run of up to RS_RUN_MAX edges of one
    (destination, relation) pair, whose source rows are summed first (include/ghf.h: ghf_run_rows_fwd)."""
    src: torch.Tensor        # [rows] int64, rows in relation order (destination ascending inside a relation); < 0: ~(row of the run sums)
    dst: torch.Tensor        # [rows] int64
    ypos: torch.Tensor       # [rows] int64: the row's position in destination order = its row of the per-row results
    slice_tab: torch.Tensor  # [S, 3] int64 (relation, first row, end row): tiles of at most RS_TILE rows
    off: torch.Tensor        # [N+1] int64: a destination's rows of the per-row results
    rows: int = 0            # number of rows (= edges without runs)
    hub_of: Optional[torch.Tensor] = None      # [N] int32: hub index or -1 (None: no destination has more than RS_HUB_ROWS rows)
    hub_tab: Optional[torch.Tensor] = None     # [H, 2] int64: (first slot, slots) of a hub's chunk sums
    hub_chunks: Optional[torch.Tensor] = None  # [C, 3] int64: (first row, end row, slot)
    cnt: Optional[torch.Tensor] = None         # [rows] float32: edges a row stands for (runs only)
    run_src: Optional[torch.Tensor] = None     # [M] int64: the sources of the runs of two or more edges, run after run
    run_start: Optional[torch.Tensor] = None   # [X+1] int64: run i sums run_src[run_start[i] : run_start[i+1]]
    deg_of: Optional[torch.Tensor] = None      # [N] int32: in-degrees (the mean's divisor when rows are runs)
    twin: Optional[object] = None              # () -> the per-edge plan of the same graph (the exact kernels have no run rows)
    _Y: Optional[torch.Tensor] = None
    _P: Optional[torch.Tensor] = None
    _X: Optional[torch.Tensor] = None
    _twin: Optional["RsPlan"] = None

    def hub_scratch(self, d: int) -> torch.Tensor:
        n = self.hub_chunks.size(0) * d
        if self._P is None or self._P.numel() < n:
            self._P = torch.empty(n, dtype=torch.float32, device=self.off.device)
        return self._P

    def scratch(self, E: int, d: int, device) -> torch.Tensor:
        """The per-row results [rows, d] (E: the caller's edge count — what `rows` is without runs)."""
        n = self.rows or E
        if self._Y is None or self._Y.numel() < n * d:
            self._Y = torch.empty(max(n, 1) * d, dtype=torch.float32, device=device)
        return self._Y

    def run_scratch(self, nruns: int, d: int) -> torch.Tensor:
        n = _native.load().ghf_split_rows_bytes(nruns, d, _native.WLAYOUT_SPLIT2H)
        if self._X is None or self._X.numel() < n:
            self._X = torch.empty(n, dtype=torch.uint8, device=self.off.device)
        return self._X

    def per_edge(self) -> "RsPlan":
        if self._twin is None:
            self._twin = self.twin()
        return self._twin


RS_TILE = 128
RS_RUN_MAX = 256            # edges per row of a run plan: longer (destination, relation) runs are cut (a wave sums a row's sources)
RS_RUNS_MAX_SHARE = 0.75    # runs pay when they leave at most this share of the rows (they cost one more pass over the sources)
RS_HUB_ROWS = 4096           # a destination with more rows than this is summed in chunks of this many by whole workgroups
SRC_MASK = 0x0FFFFFFF
# d = 64: the two-fp16-piece kernel (message_bx<64>) is 1.4x the exact fp32-MFMA kernel per edge, but its forward carries the
# range guard's read of a device word (ghf.h: ghf_set_range_flag) — a host sync, which a forward of a few hundred microseconds
# (bound by the host's launches) feels.  With the read at the END of the forward BASELINE config 2 (1 M edges) went from
# 0.67 to 0.81 ms on the faster kernel; read before the last layer (_native.RangeFlagRead) it is 0.65 ms against 0.66.  Below
# this many edges — graphs whose whole forward is a handful of launch latencies — the exact kernel (no guard) stays the default.
D64_PIECES_MIN_EDGES = 500_000


CSR_CONFIG = (1, _native.WLAYOUT_NATURAL, 0, 0)      # no destination blocks: relation-stationary layer or generic kernel


def block_kernel_max_nodes(d: int, wlayout: int) -> int:
    """How many rows of h a destination-block kernel can address: their gathers use 32-bit byte offsets into the row table
    (csrc/message_bx.hip: N (4d + 4) bytes of split rows and scales; message_pp.hip: N 4d bytes of fp32
    rows), 4 GiB less the page the kernels point dead rows at.  d = 128: 8.3 M rows."""
    per_row = 4 * d + (4 if wlayout in _native.SPLIT_LAYOUTS else 0)
    return ((1 << 32) - 4096) // per_row


def plan_config(d: int, E: int, N: Optional[int] = None) -> Tuple[int, int, int, int]:
    """(block_nodes, weight layout, chunk_rows, split_chunks) build_plan uses for a graph of E edges and N nodes (the whole
    graph's counts: every rank of a sharded run holds all rows of h): ghf_message_config, but small hidden-64 graphs keep the
    exact kernel, and graphs with more rows than the block kernels' 32-bit row offsets reach (block_kernel_max_nodes) get a
    CSR plan — the relation-stationary layer (d % 128 == 0) or the generic kernel, both on 64-bit row indices."""
    cfg = _native.message_config(d)
    if d == 64 and cfg[1] == _native.WLAYOUT_SPLIT2H and not os.environ.get("GHF_KERNEL") and E < D64_PIECES_MIN_EDGES:
        cfg = _native.exact_config(d)                    # see D64_PIECES_MIN_EDGES
    if N is not None and cfg[0] > 1 and N > block_kernel_max_nodes(d, cfg[1]):
        cfg = CSR_CONFIG
    return cfg


def exact_plan(plan: "GraphPlan", d: int) -> "GraphPlan":
    """`plan`'s edges planned for the exact kernels (built once, kept on the plan)."""
    if plan.exact is None:
        src, dst, rel = plan.edge_arrays()
        plan.exact = build_plan(torch.stack([src, dst]), rel, plan.unique_texts, plan.N, d, plan.sorted_key.device, exact=True)
        plan.exact.row_lo, plan.exact.row_hi = plan.row_lo, plan.row_hi
        plan.exact.force_exact = True                    # (wide rows: the relation-stationary layer on fp32 MFMAs)
    return plan.exact


def _hub_tables(rs: RsPlan, rows_of: torch.Tensor, N: int, dev) -> None:
    """Destinations with more than RS_HUB_ROWS rows: their rows are summed in chunks first (segment_partial)."""
    hubs = torch.nonzero(rows_of > RS_HUB_ROWS).flatten()
    if not hubs.numel():
        return
    hub_nodes = hubs.cpu().tolist()
    starts = rs.off.index_select(0, hubs).cpu().tolist()
    ends = rs.off.index_select(0, hubs + 1).cpu().tolist()
    chunks, htab = [], []
    for a, b_ in zip(starts, ends):
        htab.append((len(chunks), -(-(b_ - a) // RS_HUB_ROWS)))
        chunks += [(p, min(p + RS_HUB_ROWS, b_), len(chunks) + i) for i, p in enumerate(range(a, b_, RS_HUB_ROWS))]
    hub_of = torch.full((N,), -1, dtype=torch.int32)
    hub_of[torch.tensor(hub_nodes)] = torch.arange(len(hub_nodes), dtype=torch.int32)
    rs.hub_of, rs.hub_tab = hub_of.to(dev), torch.tensor(htab, dtype=torch.int64).to(dev)
    rs.hub_chunks = torch.tensor(chunks, dtype=torch.int64).to(dev)


def _slices(counts: List[int]) -> List[Tuple[int, int, int]]:
    # (Relation-major: the tiles in flight share their relation's 512 KB of weights in L2.  Launched band by band instead —
    # every relation's tiles of the same destination rows together, as ghf_edge_outer's slices are — the destination rows would
    # be shared, but the weights of all relations no longer fit L2: C5 layer 39.0 -> 43.9 / 40.6 / 39.4 ms with bands of 4 / 16 /
    # 64 tiles, round 4.)
    tab, e = [], 0
    for r, c in enumerate(counts):
        tab += [(r, a, min(a + RS_TILE, e + c)) for a in range(e, e + c, RS_TILE)]
        e += c
    return tab


def build_rs(plan: "GraphPlan", runs: Optional[bool] = None) -> RsPlan:
    """From a CSR plan (block_nodes == 1: edges sorted by key = dst * R + relation): the same edges grouped by relation.
    One device sort and a few host syncs per plan.  runs: rows = runs of equal (destination, relation) (None: when that
    leaves at most RS_RUNS_MAX_SHARE of the rows — power-law graphs: a hub's edges repeat its relations; GHF_RS_RUNS=0/1
    forces)."""
    if plan.block_nodes != 1:
        raise ValueError("the relation-stationary layer runs on CSR plans (block_nodes == 1)")
    dev, R, N, E = plan.sorted_key.device, plan.R, plan.N, plan.E
    if E == 0:
        z = torch.zeros(1, dtype=torch.int64, device=dev)
        return RsPlan(src=z, dst=z, ypos=z, slice_tab=torch.zeros(0, 3, dtype=torch.int64, device=dev),
                      off=torch.zeros(N + 1, dtype=torch.int64, device=dev))
    key = plan.sorted_key[:E].to(torch.int64) & 0xFFFFFFFF
    src = plan.sorted_src[:E].to(torch.int64) & SRC_MASK
    if os.environ.get("GHF_RS_RUNS") in ("0", "1"):
        runs = os.environ["GHF_RS_RUNS"] == "1"
    if runs is None or runs:
        head = torch.ones(E, dtype=torch.bool, device=dev)
        head[1:] = key[1:] != key[:-1]
        lo = torch.nonzero(head).flatten()                            # first edge of every (dst, rel) run
        n = torch.diff(lo, append=torch.tensor([E], device=dev))
        per = torch.div(n + (RS_RUN_MAX - 1), RS_RUN_MAX, rounding_mode="floor")      # rows of a run (long runs are cut)
        rows = int(per.sum().item())
        if runs or rows <= RS_RUNS_MAX_SHARE * E:
            return _build_rs_runs(plan, key, src, lo, n, per, rows)
    off = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(plan.indeg.to(torch.int64), 0)
    dst, rel = torch.div(key, R, rounding_mode="floor"), key % R
    perm = torch.sort(rel, stable=True).indices
    counts = torch.bincount(rel, minlength=R).cpu().tolist()
    rs = RsPlan(src=src.index_select(0, perm).contiguous(), dst=dst.index_select(0, perm).contiguous(), ypos=perm.contiguous(),
                slice_tab=torch.tensor(_slices(counts), dtype=torch.int64).to(dev), off=off, rows=E)
    _hub_tables(rs, plan.indeg, N, dev)
    return rs


def _build_rs_runs(plan: "GraphPlan", key, src, lo, n, per, rows: int) -> RsPlan:
    dev, R, N, E = key.device, plan.R, plan.N, plan.E
    first = torch.cumsum(per, 0) - per                                 # first row of every run
    run_of = torch.repeat_interleave(torch.arange(lo.numel(), device=dev), per, output_size=rows)
    j = torch.arange(rows, device=dev) - first.index_select(0, run_of)
    r_lo = lo.index_select(0, run_of) + RS_RUN_MAX * j                  # a row's edges: [r_lo, r_lo + r_n) in CSR order
    r_n = torch.minimum(n.index_select(0, run_of) - RS_RUN_MAX * j, torch.tensor(RS_RUN_MAX, device=dev))
    k0 = key.index_select(0, r_lo)
    r_dst, r_rel = torch.div(k0, R, rounding_mode="floor"), k0 % R
    multi = r_n > 1
    x_of = torch.cumsum(multi.to(torch.int64), 0) - 1                   # row of the run sums, for rows of two or more edges
    code = torch.where(multi, -x_of - 1, src.index_select(0, r_lo))
    m_lo, m_n = r_lo[multi], r_n[multi]
    run_start = torch.zeros(m_n.numel() + 1, dtype=torch.int64, device=dev)
    run_start[1:] = torch.cumsum(m_n, 0)
    M = int(run_start[-1].item())
    which = torch.repeat_interleave(torch.arange(m_n.numel(), device=dev), m_n, output_size=M)
    run_src = src.index_select(0, torch.arange(M, device=dev) - run_start.index_select(0, which) + m_lo.index_select(0, which))
    perm = torch.sort(r_rel, stable=True).indices                       # rows are in destination order: ascending inside a relation
    counts = torch.bincount(r_rel, minlength=R).cpu().tolist()
    rows_of = torch.bincount(r_dst, minlength=N)
    off = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(rows_of, 0)
    rs = RsPlan(src=code.index_select(0, perm).contiguous(), dst=r_dst.index_select(0, perm).contiguous(), ypos=perm.contiguous(),
                slice_tab=torch.tensor(_slices(counts), dtype=torch.int64).to(dev), off=off, rows=rows,
                cnt=r_n.index_select(0, perm).to(torch.float32).contiguous(), run_src=run_src.contiguous(), run_start=run_start,
                deg_of=plan.indeg, twin=lambda: build_rs(plan, runs=False))
    _hub_tables(rs, rows_of, N, dev)
    return rs


@dataclass
class GraphPlan:
    N: int
    E: int
    R: int
    block_nodes: int
    wlayout: int
    unique_texts: List[str]
    rel_ids: torch.Tensor          # [E] int64, device
    sorted_key: torch.Tensor       # [E] uint32 bit patterns in an int32 tensor
    sorted_src: torch.Tensor       # [E] int32
    seg_off: torch.Tensor          # [nseg+1] int32
    indeg: torch.Tensor            # [N] int32
    chunk_tab: Optional[torch.Tensor] = None       # [2*max_chunks] int32 (block plans)
    blk_chunk_off: Optional[torch.Tensor] = None   # [NB+1] int32 (block plans)
    item_tab: Optional[torch.Tensor] = None        # [4*max_items] int32 work items (block plans)
    blk_item_off: Optional[torch.Tensor] = None    # [NB+1] int32
    item_off_host: Optional[np.ndarray] = None     # host copy of blk_item_off (launch geometry)
    n_slots: int = 0                               # scratch slots (BN*d floats each) the split blocks need
    _partial: Optional[torch.Tensor] = None
    chunk_rows: int = 0
    row_lo: int = 0                # destination rows this plan covers (multi-GPU shards)
    row_hi: int = 0
    train: Optional[object] = None  # autograd.TrainPlan, built by the first forward that records gradients
    rs: Optional[RsPlan] = None     # relation-stationary extras, built by the first wide-row forward
    exact: Optional["GraphPlan"] = None   # the same edges planned for the exact fp32 kernels (range guard fallback)
    force_exact: bool = False             # this plan's wide-row layer runs on fp32 MFMAs (exact plans: the range guard's fallback)

    def edge_arrays(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(src, dst, relation id) int64 [E] of this plan's edges, decoded from the sorted arrays."""
        key = self.sorted_key[: self.E].to(torch.int64) & 0xFFFFFFFF
        raw = self.sorted_src[: self.E].to(torch.int64)
        if self.block_nodes == 1:
            return raw & 0xFFFFFFFF, torch.div(key, self.R, rounding_mode="floor"), key % self.R
        bn = self.block_nodes
        blk, rem = torch.div(key, self.R * bn, rounding_mode="floor"), key % (self.R * bn)
        return raw & SRC_MASK, blk * bn + rem % bn, torch.div(rem, bn, rounding_mode="floor")

    def bytes(self) -> int:
        ts = (self.rel_ids, self.sorted_key, self.sorted_src, self.seg_off, self.indeg, self.chunk_tab, self.blk_chunk_off,
              self.item_tab, self.blk_item_off)
        return sum(t.numel() * t.element_size() for t in ts if t is not None)

    def items_for(self, row0: int, rows: int, d: int):
        """(first work item, item count, scratch) for the destination rows [row0, row0+rows) of a block plan."""
        if self.block_nodes == 1:
            return 0, 0, None
        bn = self.block_nodes
        i0, i1 = int(self.item_off_host[row0 // bn]), int(self.item_off_host[-(-(row0 + rows) // bn)])
        if self.n_slots and (self._partial is None or self._partial.numel() < self.n_slots * bn * d):
            self._partial = torch.empty(self.n_slots * bn * d, dtype=torch.float32, device=self.sorted_key.device)
        return i0, i1 - i0, (self._partial if self.n_slots else None)


def build_plan(edge_index: torch.Tensor, rel_ids: torch.Tensor, unique_texts: List[str], N: int, d: int,
               device: torch.device, force_generic: bool = False,
               row_range: Optional[Tuple[int, int]] = None,
               owner: Optional[Tuple[int, int, int]] = None,
               owner_bounds: Optional[Tuple[torch.Tensor, int, int]] = None,
               edge_range: Optional[Tuple[int, int]] = None, exact: bool = False) -> GraphPlan:
    """Run K0 on `device`.  Raises IndexError on out-of-range node or relation ids.

    Multi-GPU shards keep only the in-edges of the rows they own: `row_range=(lo, hi)` for one contiguous range,
    `owner=(S, G, g)` for block-cyclic ownership (row v belongs to rank (v // S) % G), or `owner_bounds=(bounds, G, g)`
    for slots of unequal size (row v lies in slot s: bounds[s] <= v < bounds[s+1]; slot s belongs to rank s % G — shards
    balanced by in-edge count).  `edge_range=(lo, hi)` keeps the edges with these positions in the caller's list whatever
    their ends (edge-range shards: the ranks' partial sums are reduced afterwards)."""
    if edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise ValueError(f"edge_index must be [2, E], got {tuple(edge_index.shape)}")
    E = edge_index.size(1)
    if E == 0:
        raise ValueError("edge_index has no edges; the reference cannot encode an empty relation list either")
    R = len(unique_texts)
    if exact:                                     # the exact fp32 kernels (range guard fallback: _native.exact_config)
        bn, wl, cr, sc = _native.exact_config(d)
        if bn > 1 and N > block_kernel_max_nodes(d, wl):
            bn, wl, cr, sc = CSR_CONFIG
    else:
        bn, wl, cr, sc = CSR_CONFIG if force_generic else plan_config(d, edge_index.size(1), N)
    ei = edge_index.to(device=device, dtype=torch.int64).contiguous()
    rel = rel_ids.to(device=device, dtype=torch.int64).contiguous()
    lo, hi = (0, N) if row_range is None else row_range
    if row_range is not None or owner is not None or owner_bounds is not None or edge_range is not None:
        if edge_range is not None:
            keep = torch.zeros(ei.size(1), dtype=torch.bool, device=device)
            keep[edge_range[0]:edge_range[1]] = True
        elif owner_bounds is not None:
            bounds, G, g = owner_bounds
            slot = torch.bucketize(ei[1], bounds.to(device=device, dtype=torch.int64)[1:], right=True)
            keep = (slot % G) == g
        elif owner is not None:
            S, G, g = owner
            keep = (torch.div(ei[1], S, rounding_mode="floor") % G) == g
        else:
            keep = (ei[1] >= lo) & (ei[1] < hi)
        ei = ei[:, keep].contiguous()
        rel = rel[keep].contiguous()
        if ei.size(1) == 0:                      # a shard without in-edges: empty plan, every row is "isolated"
            nseg = N if bn == 1 else ((N + bn - 1) // bn) * R
            z = lambda n: torch.zeros(n, dtype=torch.int32, device=device)  # noqa: E731
            nb = (N + bn - 1) // bn
            empty = GraphPlan(N=N, E=0, R=R, block_nodes=bn, wlayout=wl, unique_texts=unique_texts, rel_ids=rel,
                              sorted_key=z(1), sorted_src=z(1), seg_off=z(nseg + 1), indeg=z(N), chunk_rows=cr,
                              row_lo=lo, row_hi=hi)
            if bn > 1:                                # one empty work item per block
                items = np.zeros((nb, 4), dtype=np.int32)
                items[:, 0], items[:, 3] = np.arange(nb), -1
                empty.chunk_tab, empty.blk_chunk_off = z(2), z(nb + 1)
                empty.item_tab = torch.from_numpy(items.reshape(-1)).to(device)
                empty.item_off_host = np.arange(nb + 1, dtype=np.int32)
                empty.blk_item_off = torch.from_numpy(empty.item_off_host).to(device)
            return empty
    pl = _native.plan_build(ei, rel, N, R, bn, cr, sc)
    status = pl["status"].cpu().numpy()           # the only host sync of the plan
    st = int(status[0])
    if st & 1:
        raise IndexError(f"edge_index holds node ids outside [0, {N})")
    if st & 2:
        raise IndexError(f"relation ids outside [0, {R})")
    plan = GraphPlan(N=N, E=ei.size(1), R=R, block_nodes=bn, wlayout=wl, unique_texts=unique_texts, rel_ids=rel,
                     sorted_key=pl["sorted_key"], sorted_src=pl["sorted_src"], seg_off=pl["seg_off"], indeg=pl["indeg"],
                     chunk_tab=pl["chunk_tab"], blk_chunk_off=pl["blk_chunk_off"], item_tab=pl["item_tab"],
                     blk_item_off=pl["blk_item_off"], chunk_rows=cr, row_lo=lo, row_hi=hi)
    if bn > 1:
        plan.item_off_host = pl["blk_item_off"].cpu().numpy()
        plan.n_slots = int(status[2])
    return plan


FULL_FINGERPRINT_MAX = 1 << 17     # lists up to this length are fingerprinted whole
SAMPLED_POSITIONS = 4096            # longer ones at this many seeded-random positions (plus both ends): an early miss only —
_SAMPLE_IDX: dict = {}              # a hit on a longer list is then verified entry by entry (same_relations)


def _texts_fingerprint(edge_texts: Sequence[str]) -> Tuple:
    """What the plan cache's KEY holds of a relation list besides its identity and length.

    The reference maps the strings to ids on every call (models/hypergnn.py:264-268), so a list edited in place must not
    hit a stale plan.  Up to FULL_FINGERPRINT_MAX entries the whole content is hashed (strings cache their hashes: ~1 ms
    at 2^17).  A longer list is sampled at SAMPLED_POSITIONS positions drawn once per length from a seeded generator: that
    only turns most edits into an immediate miss — a hit is confirmed against a snapshot of the whole list
    (`PlanCache.verifier`, `same_relations`) before its result is returned."""
    n = len(edge_texts)
    if n == 0:
        return (0,)
    if n <= FULL_FINGERPRINT_MAX:
        return (n, hash(tuple(edge_texts)))
    pick = _SAMPLE_IDX.get(n)
    if pick is None:
        idx = np.unique(np.concatenate([np.random.default_rng(n).integers(0, n, SAMPLED_POSITIONS), [0, n - 1]])).tolist()
        pick = (idx, operator.itemgetter(*idx))   # (one C call for the 4 k lookups: 45 us where a Python loop took 160)
        if len(_SAMPLE_IDX) >= 16:
            _SAMPLE_IDX.pop(next(iter(_SAMPLE_IDX)))
        _SAMPLE_IDX[n] = pick
    return (n, hash(pick[1](edge_texts)))


# ---- a long relation list against the snapshot taken when its plan was built ------------------------------------------
# The snapshot is a shallow copy of the list — it keeps every original string object alive, so no address can be reused and
# "same pointer" means "same (immutable) string" — plus checksums of its array of object pointers.  CPython keeps a list's
# items as one array of pointers (PyListObject.ob_item, behind ob_refcnt / ob_type / ob_size; a tuple's items follow its header
# directly); a hit is confirmed by checksumming the LIVE list's array (ghf_host_checksum64 over a few threads — ctypes releases
# the GIL; 80 MB at 10 M edges, ~4 ms, memory-bound) and comparing with the snapshot's: an unedited list never costs more.
# A mismatch falls back to comparing the strings themselves.
_VERIFY_THREADS = max(1, min(8, (os.cpu_count() or 4)))
_VERIFY_MIN_PARALLEL = 1 << 20
_verify_pool = None


def _pool():
    global _verify_pool
    if _verify_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _verify_pool = ThreadPoolExecutor(max_workers=_VERIFY_THREADS, thread_name_prefix="ghf-plan-verify")
    return _verify_pool


_check_pool = None


def check_pool():
    """Where HyperGNN.forward runs a cache hit's whole-list check beside its launches (apart from `_pool`, whose workers
    the check itself waits on)."""
    global _check_pool
    if _check_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _check_pool = ThreadPoolExecutor(max_workers=2, thread_name_prefix="ghf-plan-check")
    return _check_pool


def _item_array_address(seq) -> Optional[int]:
    if sys.implementation.name != "cpython":
        return None
    if type(seq) is list:
        return ctypes.c_void_p.from_address(id(seq) + 3 * ctypes.sizeof(ctypes.c_void_p)).value
    if type(seq) is tuple:
        return id(seq) + 3 * ctypes.sizeof(ctypes.c_void_p)
    return None


def pointer_checksums(seq: Sequence) -> Optional[Tuple[int, ...]]:
    """Checksums of the object pointers `seq` holds (its identity, entry by entry), one per slice of the list; None where the
    item array cannot be addressed (not CPython, not a list / tuple)."""
    n = len(seq)
    base = _item_array_address(seq)
    if base is None:
        return None
    if n == 0:
        return (0,)
    word = ctypes.sizeof(ctypes.c_void_p)
    fn = _native.load().ghf_host_checksum64
    if n < _VERIFY_MIN_PARALLEL:
        return (n, fn(base, n * word, 0))
    step = -(-n // _VERIFY_THREADS)
    jobs = [_pool().submit(fn, base + i * word, (min(i + step, n) - i) * word, i) for i in range(0, n, step)]
    return (n,) + tuple(j.result() for j in jobs)


def same_objects(a: Sequence, b: Sequence) -> bool:
    """len(a) == len(b) and a[i] is b[i] for every i."""
    if len(a) != len(b):
        return False
    ca, cb = pointer_checksums(a), pointer_checksums(b)
    if ca is None or cb is None:
        return all(map(operator.is_, a, b))
    return ca == cb


def same_relations(edge_texts: Sequence[str], snapshot: Sequence[str]) -> bool:
    """Does `edge_texts` still spell the relations its plan was built for?  Same objects (the fast path: an unedited list),
    else equal strings (a list refilled with equal strings maps to the same ids: reference models/hypergnn.py:264-268)."""
    if same_objects(edge_texts, snapshot):
        return True
    if len(edge_texts) != len(snapshot):
        return False
    return (edge_texts if type(edge_texts) is list else list(edge_texts)) == (snapshot if type(snapshot) is list else list(snapshot))


class PlanCache:
    """Small LRU of GraphPlans keyed on the identity of the inputs.

    Key: edge_index storage pointer, shape, in-place version counter and device;
    the edge_texts list object identity and its content fingerprint
    (`_texts_fingerprint`: the whole list up to 2^17 entries, 4096 seeded-random
    positions beyond); N; d.  The cache keeps references to both inputs so their
    ids cannot be recycled while an entry lives.  A hit on a list longer than
    2^17 entries is exact too: the entry holds a snapshot (shallow copy) of the
    list and `verifier(key, edge_texts)` returns the check the caller runs
    before it trusts the hit — `HyperGNN.forward` runs it on the host while the
    GPU computes on the cached plan, and reruns on a fresh plan if it fails —
    so an in-place edit anywhere is followed, as in the reference
    (models/hypergnn.py:264-268).  Writes through `edge_index.data` bypass the
    version counter and are not seen.  `GHF_PLAN_CACHE=0` makes every lookup a miss.
    """

    def __init__(self, capacity: int = 4) -> None:
        self.capacity = capacity
        self._entries: "OrderedDict[Tuple, Tuple[GraphPlan, object, object, object]]" = OrderedDict()
        self.hits = 0
        self.misses = 0
        self.stale = 0                 # hits that the whole-list check turned into misses

    @staticmethod
    def key(edge_index: torch.Tensor, edge_texts: Sequence[str], N: int, d: int, device: torch.device,
            extra: Tuple = ()) -> Tuple:
        return (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, str(edge_index.device),
                id(edge_texts), _texts_fingerprint(edge_texts), N, d, str(device)) + tuple(extra)

    def get(self, key: Tuple) -> Optional[GraphPlan]:
        if os.environ.get("GHF_PLAN_CACHE") == "0":
            self.misses += 1
            return None
        ent = self._entries.get(key)
        if ent is None:
            self.misses += 1
            return None
        self._entries.move_to_end(key)
        self.hits += 1
        return ent[0]

    def put(self, key: Tuple, plan: GraphPlan, edge_index: torch.Tensor, edge_texts: Sequence[str], objects=None,
            background: bool = False):
        """objects: the list's distinct string objects when relation_ids mapped it by identity (want_objects).  Holding THEM
        keeps every address of the list's pointer array taken, so the checksums of that array are all a later hit needs: no
        shallow copy of ten million pointers (19 -> 0 ms of a cold forward at C3).  Without them (strings hashed by value: every
        entry may be its own object) the entry keeps the copy, as before.
        background: take the checksums on the cache's threads (GIL-free) and return the future; the caller collects it before
        it hands anything back (HyperGNN.forward: beside its launches)."""
        # (a list the key's fingerprint covers whole needs no snapshot; forward_ids hands over (ids tensor, texts): no list)
        long_list = isinstance(edge_texts, (list, tuple)) and len(edge_texts) > FULL_FINGERPRINT_MAX and \
            not (len(edge_texts) == 2 and isinstance(edge_texts[0], torch.Tensor))
        snap, fut = None, None
        if long_list and objects is not None and _item_array_address(edge_texts) is not None:
            snap = [None, None, objects]
            if background:
                def take(snap=snap, texts=edge_texts):
                    snap[1] = pointer_checksums(texts)
                fut = check_pool().submit(take)
            else:
                snap[1] = pointer_checksums(edge_texts)
        elif long_list:
            copy = list(edge_texts)
            snap = [copy, pointer_checksums(copy), None]
        self._entries[key] = (plan, edge_index, edge_texts, snap)
        self._entries.move_to_end(key)
        while len(self._entries) > self.capacity:
            self._entries.popitem(last=False)
        return fut

    def verifier(self, key: Tuple, edge_texts: Sequence[str]):
        """None when a hit on `key` needs no further check; else a callable () -> bool (thread-safe, GIL-free for most of its
        time): True when the whole list still spells the plan's relations (a list refilled with equal strings becomes the new
        snapshot); False drops the entry."""
        ent = self._entries.get(key)
        if ent is None or ent[3] is None:
            return None

        def check() -> bool:
            snap = ent[3]
            live = pointer_checksums(edge_texts)
            if live is not None and snap[1] is not None and live == snap[1]:
                return True
            if snap[0] is not None:                        # (entries that keep a copy of the list: compare with it)
                if live is None and same_objects(edge_texts, snap[0]):
                    return True
                if same_relations(edge_texts, snap[0]):
                    copy = list(edge_texts)
                    snap[0], snap[1] = copy, pointer_checksums(copy)
                    return True
            # (entries that hold the distinct objects only: some entry now points elsewhere — an edit, or equal strings in
            # new objects; either way the caller maps the list again, which costs what comparing it would)
            if self._entries.get(key) is ent:
                del self._entries[key]
            self.hits -= 1
            self.misses += 1
            self.stale += 1
            return False
        return check

    def clear(self) -> None:
        self._entries.clear()

    def __len__(self) -> int:
        return len(self._entries)
