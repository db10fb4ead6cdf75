"""Graph plan: relation ids + the sorted edge arrays the HIP kernels consume, and its cache.

Replaces the per-forward host work of the reference (models/hypergnn.py:264-268:
dict.fromkeys dedupe, id list, torch.tensor) with a plan that is built once per
(edge_index, edge_texts) pair and reused: the `List[str]` boundary costs ~1.5 s
at 10 M edges, far more than the whole device forward.
"""

from __future__ import annotations

import os
from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native


def relation_ids(edge_texts: Sequence[str]) -> Tuple[List[str], np.ndarray]:
    """Unique relation strings in first-appearance order and the per-edge id (int64).

    Same mapping as the reference (models/hypergnn.py:264-268); the order of the
    unique list only permutes the generated weights, never the result.
    """
    unique = list(dict.fromkeys(edge_texts))
    lut = {t: i for i, t in enumerate(unique)}
    ids = np.fromiter(map(lut.__getitem__, edge_texts), dtype=np.int64, count=len(edge_texts))
    return unique, ids


@dataclass
class RsPlan:
    """What the relation-stationary layer (csrc/message_rs.hip, wide hidden sizes) reads besides the CSR plan."""
    src: torch.Tensor        # [E] int64, edges in relation order (destination ascending inside a relation)
    dst: torch.Tensor        # [E] int64
    ypos: torch.Tensor       # [E] int64: the edge's position in destination order = its row of the per-edge results
    slice_tab: torch.Tensor  # [S, 3] int64 (relation, first edge, end edge): tiles of at most RS_TILE edges
    off: torch.Tensor        # [N+1] int64: a destination's rows of the per-edge results
    hub_of: Optional[torch.Tensor] = None      # [N] int32: hub index or -1 (None: no destination has more than RS_HUB_ROWS rows)
    hub_tab: Optional[torch.Tensor] = None     # [H, 2] int64: (first slot, slots) of a hub's chunk sums
    hub_chunks: Optional[torch.Tensor] = None  # [C, 3] int64: (first row, end row, slot)
    _Y: Optional[torch.Tensor] = None
    _P: Optional[torch.Tensor] = None

    def hub_scratch(self, d: int) -> torch.Tensor:
        n = self.hub_chunks.size(0) * d
        if self._P is None or self._P.numel() < n:
            self._P = torch.empty(n, dtype=torch.float32, device=self.off.device)
        return self._P

    def scratch(self, E: int, d: int, device) -> torch.Tensor:
        if self._Y is None or self._Y.numel() < E * d:
            self._Y = torch.empty(max(E, 1) * d, dtype=torch.float32, device=device)
        return self._Y


RS_TILE = 128
RS_HUB_ROWS = 4096           # a destination with more rows than this is summed in chunks of this many by whole workgroups
SRC_MASK = 0x0FFFFFFF
# d = 64: the two-fp16-piece kernel (message_bx<64>) is 1.4x the exact fp32-MFMA kernel per edge, but its forward ends with the
# range guard's read of a device word (ghf.h: ghf_set_range_flag) — a host sync.  A forward of a few hundred microseconds is
# bound by the host's launches, which that sync stops from running ahead: at BASELINE config 2 (1 M edges) the kernel went
# 0.237 -> 0.207 ms and the forward 0.67 -> 0.81 ms.  Below this many edges the exact kernel (no guard) is the default.
D64_PIECES_MIN_EDGES = 4_000_000


def plan_config(d: int, E: int) -> Tuple[int, int, int, int]:
    """(block_nodes, weight layout, chunk_rows, split_chunks) build_plan uses for a graph of E edges (all of them: every rank
    of a sharded run passes the whole graph's count): ghf_message_config, but small hidden-64 graphs keep the exact kernel."""
    cfg = _native.message_config(d)
    if d == 64 and cfg[1] == _native.WLAYOUT_SPLIT2H and not os.environ.get("GHF_KERNEL") and E < D64_PIECES_MIN_EDGES:
        cfg = _native.exact_config(d)                    # see D64_PIECES_MIN_EDGES
    return cfg


def exact_plan(plan: "GraphPlan", d: int) -> "GraphPlan":
    """`plan`'s edges planned for the exact kernels (built once, kept on the plan)."""
    if plan.exact is None:
        src, dst, rel = plan.edge_arrays()
        plan.exact = build_plan(torch.stack([src, dst]), rel, plan.unique_texts, plan.N, d, plan.sorted_key.device, exact=True)
        plan.exact.row_lo, plan.exact.row_hi = plan.row_lo, plan.row_hi
    return plan.exact


def build_rs(plan: "GraphPlan") -> RsPlan:
    """From a CSR plan (block_nodes == 1: edges sorted by key = dst * R + relation): the same edges grouped by relation.
    One device sort and one host sync per plan."""
    if plan.block_nodes != 1:
        raise ValueError("the relation-stationary layer runs on CSR plans (block_nodes == 1)")
    dev, R, N, E = plan.sorted_key.device, plan.R, plan.N, plan.E
    off = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(plan.indeg.to(torch.int64), 0)
    if E == 0:
        z = torch.zeros(1, dtype=torch.int64, device=dev)
        return RsPlan(src=z, dst=z, ypos=z, slice_tab=torch.zeros(0, 3, dtype=torch.int64, device=dev), off=off)
    key = plan.sorted_key[:E].to(torch.int64) & 0xFFFFFFFF
    dst, rel = torch.div(key, R, rounding_mode="floor"), key % R
    src = plan.sorted_src[:E].to(torch.int64) & SRC_MASK
    perm = torch.sort(rel, stable=True).indices
    counts = torch.bincount(rel, minlength=R).cpu().tolist()
    tab, e = [], 0
    for r, c in enumerate(counts):
        tab += [(r, a, min(a + RS_TILE, e + c)) for a in range(e, e + c, RS_TILE)]
        e += c
    rs = RsPlan(src=src.index_select(0, perm).contiguous(), dst=dst.index_select(0, perm).contiguous(), ypos=perm.contiguous(),
                slice_tab=torch.tensor(tab, dtype=torch.int64).to(dev), off=off)
    hubs = torch.nonzero(plan.indeg > RS_HUB_ROWS).flatten()
    if hubs.numel():
        hub_nodes = hubs.cpu().tolist()
        starts = off.index_select(0, hubs).cpu().tolist()
        ends = off.index_select(0, hubs + 1).cpu().tolist()
        chunks, htab = [], []
        for a, b_ in zip(starts, ends):
            htab.append((len(chunks), -(-(b_ - a) // RS_HUB_ROWS)))
            chunks += [(p, min(p + RS_HUB_ROWS, b_), len(chunks) + i) for i, p in enumerate(range(a, b_, RS_HUB_ROWS))]
        hub_of = torch.full((N,), -1, dtype=torch.int32)
        hub_of[torch.tensor(hub_nodes)] = torch.arange(len(hub_nodes), dtype=torch.int32)
        rs.hub_of, rs.hub_tab = hub_of.to(dev), torch.tensor(htab, dtype=torch.int64).to(dev)
        rs.hub_chunks = torch.tensor(chunks, dtype=torch.int64).to(dev)
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
    else:
        bn, wl, cr, sc = (1, _native.WLAYOUT_NATURAL, 0, 0) if force_generic else plan_config(d, edge_index.size(1))
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
SAMPLED_POSITIONS = 4096            # longer ones at this many seeded-random positions (plus both ends)
_SAMPLE_IDX: dict = {}


def _texts_fingerprint(edge_texts: Sequence[str]) -> Tuple:
    """What the plan cache compares of a relation list besides its identity and length.

    The reference maps the strings to ids on every call (models/hypergnn.py:264-268), so a list edited in place must not
    hit a stale plan.  Up to FULL_FINGERPRINT_MAX entries the whole content is hashed (strings cache their hashes: ~1 ms
    at 2^17); a longer list is sampled at SAMPLED_POSITIONS positions drawn once per length from a seeded generator — an
    in-place edit of a 10 M-entry list is caught with the probability that it touches a sampled position or changes
    the length (a full pass over 10 M Python objects costs several warm forwards; `GHF_PLAN_CACHE=0` disables the
    cache for callers who edit large lists in place, `clear_plan_cache()` drops it once)."""
    n = len(edge_texts)
    if n == 0:
        return (0,)
    if n <= FULL_FINGERPRINT_MAX:
        return (n, hash(tuple(edge_texts)))
    idx = _SAMPLE_IDX.get(n)
    if idx is None:
        idx = np.unique(np.concatenate([np.random.default_rng(n).integers(0, n, SAMPLED_POSITIONS), [0, n - 1]])).tolist()
        if len(_SAMPLE_IDX) >= 16:
            _SAMPLE_IDX.pop(next(iter(_SAMPLE_IDX)))
        _SAMPLE_IDX[n] = idx
    return (n, hash(tuple(edge_texts[i] for i in idx)))


class PlanCache:
    """Small LRU of GraphPlans keyed on the identity of the inputs.

    Key: edge_index storage pointer, shape, in-place version counter and device;
    the edge_texts list object identity and its content fingerprint
    (`_texts_fingerprint`: the whole list up to 2^17 entries, 4096 seeded-random
    positions beyond); N; d.  The cache keeps references to both inputs so their
    ids cannot be recycled while an entry lives.  Writes through
    `edge_index.data` bypass the version counter and are not seen.
    `GHF_PLAN_CACHE=0` makes every lookup a miss.
    """

    def __init__(self, capacity: int = 4) -> None:
        self.capacity = capacity
        self._entries: "OrderedDict[Tuple, Tuple[GraphPlan, object, object]]" = OrderedDict()
        self.hits = 0
        self.misses = 0

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

    def put(self, key: Tuple, plan: GraphPlan, edge_index: torch.Tensor, edge_texts: Sequence[str]) -> None:
        self._entries[key] = (plan, edge_index, edge_texts)
        self._entries.move_to_end(key)
        while len(self._entries) > self.capacity:
            self._entries.popitem(last=False)

    def clear(self) -> None:
        self._entries.clear()

    def __len__(self) -> int:
        return len(self._entries)
