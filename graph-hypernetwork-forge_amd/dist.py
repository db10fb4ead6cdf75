"""Multi-GPU forward: one process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).

Two partitions of the path, behind one interface (`ShardedHyperGNN`):

* ``mode="dst"`` (default) — shards by DESTINATION node: a rank owns every in-edge of its rows, so the per-destination sums
  never cross ranks and the fused tail stays local.  The one exchange per layer makes the new rows visible everywhere: an
  all-gather of rows (half the bytes of the reduction an edge-range split needs).  What travels is what the next layer's
  kernel gathers — for the d = 128 kernel the rows already cut into their two fp16 pieces plus the row scales (written by
  the fused tail; the same volume as fp32), so no rank re-splits the full h; fp32 rows travel once, after the last layer.
  Ownership is block-cyclic so that the exchange overlaps the compute: the rows are cut into C x G slots (BN-aligned), slot
  c*G + g belongs to rank g, and a layer runs as
      for c in chunks:  message kernel on my slot of chunk c (compute stream);  exchange of chunk c (comm stream)
  so chunk c travels while chunk c + 1 computes.  Slots hold equal numbers of rows (``balance="rows"``) or about equal
  numbers of in-edges (``balance="edges"``: power-law graphs, BASELINE config 5).
  The exchange is either ``all_gather_into_tensor`` per chunk (RCCL picks ring or direct) or ``exchange="pairs"``: every
  rank sends its slot straight to each peer (batched send/recv) — xGMI is a full mesh of point-to-point links, and a
  pairwise exchange uses all seven of a GPU's links at once where a ring is bound by one.  ``exchange="sparse"`` sends
  only what the receiver reads: a rank gathers the SOURCE rows of its in-edges, so at plan time every rank tells each peer
  which of that peer's rows it needs (per chunk), and a layer's exchange is gather-pack -> pairwise send/recv -> scatter
  of exactly those rows (a uniform random shard of C3 at 8 ranks reads 71 % of the rows; graphs with locality far
  fewer).  The last layer's rows travel whole: every rank returns the full [N, d].
* ``mode="edges"`` — BASELINE.json's north-star split (config 4): every rank takes a contiguous range of the EDGE list
  (balanced by construction), computes raw partial sums for all rows (GHF_FLAG_RAW_SUM), the partial sums are reduced
  (reduce-scatter; all-reduce where the backend has no reduce-scatter), each rank divides by the global in-degree and runs
  the tail (ghf_tail_fwd) on its rows, and the new rows are all-gathered.  Twice the exchange volume of ``dst`` and
  chunks of 1/G the rows per weight fetch; kept for A/B (bench.py --dist-mode).

Weight generation (0.5 GFLOP) and the input projection (33 GFLOP) are recomputed on every rank instead of exchanged.
The compute steps come from an `ops` object so that the sharding / exchange logic runs on CPU with gloo
(tests/test_dist_gloo.py injects the oracle there); the product default, NativeOps, calls the HIP library and nothing else.
Collective errors propagate: a failed RCCL call raises on the rank that saw it and the job exits non-zero.
"""

from __future__ import annotations

import os
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _native
from .plan import CSR_CONFIG, GraphPlan, PlanCache, block_kernel_max_nodes, build_plan, build_rs, plan_config, relation_ids


@dataclass
class ShardSpec:
    N: int
    world: int
    rank: int
    block_nodes: int
    chunks: int                 # C
    bounds: List[int]           # C*G + 1 row boundaries (multiples of block_nodes); slot c*G + g = [bounds[s], bounds[s+1])
    uniform: bool = True        # every slot holds S rows (then the chunk all-gather is in place)

    @property
    def S(self) -> int:
        return self.bounds[1] - self.bounds[0]

    @property
    def padded_rows(self) -> int:
        return max(self.bounds[-1], self.N)

    def slot(self, c: int, g: Optional[int] = None) -> Tuple[int, int]:
        """Row range [lo, hi) (clipped to N) that rank g owns in chunk c."""
        g = self.rank if g is None else g
        s = c * self.world + g
        return min(self.N, self.bounds[s]), min(self.N, self.bounds[s + 1])

    def chunk_rows(self, c: int) -> Tuple[int, int]:
        """Row range of chunk c (all ranks' slots, padded): what one all-gather fills."""
        return self.bounds[c * self.world], self.bounds[(c + 1) * self.world]

    def owned(self) -> List[Tuple[int, int]]:
        return [r for r in (self.slot(c) for c in range(self.chunks)) if r[1] > r[0]]

    def owner_arg(self, device=None):
        """What plan.build_plan needs to keep this rank's in-edges."""
        if self.uniform:
            return dict(owner=(self.S, self.world, self.rank))
        return dict(owner_bounds=(torch.tensor(self.bounds, dtype=torch.int64), self.world, self.rank))


def shard_spec(N: int, block_nodes: int, world: int, rank: int, chunks: int = 1,
               block_edges: Optional[Sequence[int]] = None) -> ShardSpec:
    """Block-cyclic slots of equal row count, or — with `block_edges[b]` = in-edges of destination block b — of about equal
    in-edge count (every slot still a whole number of blocks; a slot may be empty)."""
    nb = -(-N // block_nodes)
    chunks = max(1, min(chunks, -(-nb // world)))          # no more chunks than blocks per rank
    nslots = world * chunks
    if block_edges is None:
        S = -(-nb // nslots) * block_nodes
        return ShardSpec(N=N, world=world, rank=rank, block_nodes=block_nodes, chunks=chunks,
                         bounds=[s * S for s in range(nslots + 1)])
    import numpy as np
    be = np.asarray(block_edges, dtype=np.float64)
    assert be.shape[0] == nb, "one in-edge count per destination block"
    # a block's cost: its edges plus a constant per block (an empty block still runs its tail)
    cost = be + max(1.0, be.sum() / max(nb, 1) * 0.05)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    total = cum[-1]
    # chunks of equal cost; inside chunk c rank g takes blocks until its running total reaches its share of everything up
    # to and including this chunk — a rank that a hub block pushed over its share gets less (or nothing) in later chunks
    ccut = [int(np.searchsorted(cum, total * c / chunks, side="left")) for c in range(chunks + 1)]
    ccut[0], ccut[-1] = 0, nb
    cuts, have = [0], np.zeros(world)
    for c in range(chunks):
        lo, hi = max(ccut[c], cuts[-1]), max(ccut[c + 1], cuts[-1])
        b = lo
        for g in range(world):
            target = cum[hi] / world                          # rank g's share of all blocks up to the end of this chunk
            e = b
            if g == world - 1:
                e = hi
            else:
                while e < hi and have[g] + (cum[e + 1] - cum[b]) <= target + 0.5 * cost[e]:
                    e += 1
            have[g] += cum[e] - cum[b]
            cuts.append(e)
            b = e
    return ShardSpec(N=N, world=world, rank=rank, block_nodes=block_nodes, chunks=chunks,
                     bounds=[c * block_nodes for c in cuts], uniform=False)


class NativeOps:
    """The product compute steps: C-ABI calls only."""

    def message_config(self, d: int, E: int, N: Optional[int] = None, exact: bool = False):
        if exact:                                                   # the range guard's fallback (plan.build_plan(exact=True))
            cfg = _native.exact_config(d)
            return CSR_CONFIG if N is not None and cfg[0] > 1 and N > block_kernel_max_nodes(d, cfg[1]) else cfg
        return plan_config(d, E, N)

    def build_plan(self, edge_index, rel_ids, unique, N, d, device, exact: bool = False, **shard) -> GraphPlan:
        if exact:
            plan = build_plan(edge_index, rel_ids, unique, N, d, device, exact=True, **shard)
            plan.force_exact = True                      # (wide rows: pass 1 on fp32 MFMAs — carried by the plan)
            return plan
        return build_plan(edge_index, rel_ids, unique, N, d, device,
                          force_generic=_native.prefer_rs(d, len(unique)) and "edge_range" not in shard, **shard)

    def plan_sources(self, plan) -> torch.Tensor:
        """Source node of every edge this rank's plan holds (what its kernels gather from h)."""
        return plan.edge_arrays()[0]

    def text_embs(self, model, unique: Sequence[str], device) -> torch.Tensor:
        return model.text_encoder(unique, device)

    def input_proj(self, model, x: torch.Tensor, out: torch.Tensor, h_split, plan) -> None:
        """h0 for ALL rows on every rank (with the split form the first layer gathers, when the kernel wants one):
        0.7 ms of replicated compute at C3 instead of an exchange of N rows, which costs more on every fabric size
        (one xGMI link moves 258 MB in 3.4 ms at 2 GPUs; seven move 451 MB in >= 0.84 ms at 8)."""
        _native.input_proj_fwd(x, model.input_proj.weight.detach(), model.input_proj.bias.detach(), out=out,
                               h_split=h_split, split_layout=plan.wlayout if h_split is not None else 0)

    def all_weights(self, model, text_embs, plan, after=None):
        """([weights of layer l], [event l] or None): every layer's generation on a side stream (HyperGNN.generate_all)."""
        if model.num_layers <= 8 and os.environ.get("GHF_GEN_BATCHED", "1") != "0":
            return model.generate_batched(text_embs, plan.wlayout), None      # one launch sequence for all layers, on this stream
        return model.generate_all(text_embs, plan.wlayout, side_stream=plan.E >= model.SIDE_STREAM_MIN_EDGES // 8, after=after)

    def split_rows(self, plan, h):
        """What the message kernel gathers: h itself, or its rows cut into 16-bit pieces."""
        return _native.split_rows(h, plan.wlayout) if plan.wlayout in _native.SPLIT_LAYOUTS else None

    # -- exchanging the split form (SPLIT2H: N rows of 4d bytes, then N float scales) ---------------------
    def exchanges_split(self, plan) -> bool:
        return plan.wlayout == _native.WLAYOUT_SPLIT2H

    def alloc_split(self, plan, N: int, d: int, device) -> torch.Tensor:
        return _native.alloc_split(N, d, plan.wlayout, device)

    def split_parts(self, plan, hs: torch.Tensor, N: int, d: int) -> List[torch.Tensor]:
        """The row-indexed regions of a split buffer as [N, bytes] views: what the exchange moves."""
        b = hs.view(torch.uint8)
        return [b[: N * 4 * d].view(N, 4 * d), b[N * 4 * d: N * 4 * d + 4 * N].view(N, 4)]

    def pack_rows(self, tables: List[torch.Tensor], idx: torch.Tensor) -> torch.Tensor:
        """The sparse exchange's message: rows `idx` of every table of `tables` (the split form's rows and scales, or fp32
        rows), one contiguous [n, bytes] buffer — ghf_rows_pack (csrc/exchange.hip)."""
        return _native.rows_pack([t.view(torch.uint8).view(t.size(0), -1) for t in tables], idx)

    def unpack_rows(self, tables: List[torch.Tensor], idx: torch.Tensor, packed: torch.Tensor) -> None:
        _native.rows_unpack([t.view(torch.uint8).view(t.size(0), -1) for t in tables], idx, packed)

    def layer_begin(self, model, l: int, weights, h, plan) -> None:
        """Once per layer, before its chunks.  Wide rows (csrc/message_rs.hip): pass 1 over all of this rank's edges —
        it runs in relation order, not by destination chunk; the chunks then only do pass 2 on their rows."""
        if plan.block_nodes == 1 and _native.rs_supported(h.size(1)) and plan.E > 0:
            if plan.rs is None:
                plan.rs = build_rs(plan)
            W_msg, W_self, bias = weights
            _native.edge_transform_fwd(h, plan.rs, W_msg, W_self, bias, plan.rs.scratch(plan.E, h.size(1), h.device), exact=plan.force_exact)

    def layer_rows(self, model, l: int, weights, h, h_split, plan, h_out, lo: int, hi: int, h_split_out=None) -> None:
        norm = model.layer_norms[l]
        W, W_self, bias = weights
        if plan.block_nodes == 1 and _native.rs_supported(h.size(1)):
            if plan.rs is None:
                plan.rs = build_rs(plan)
            _native.segment_tail_fwd(plan.rs.scratch(plan.E, h.size(1), h.device), plan.rs, h, norm.weight.detach(),
                                     norm.bias.detach(), norm.eps, h_out, row0=lo, rows=hi - lo, exact=plan.force_exact)
            return
        _native.message_layer_fwd(h, plan, W, W_self, bias, plan.wlayout, norm.weight.detach(), norm.bias.detach(),
                                  norm.eps, h_out, row0=lo, rows=hi - lo, h_split=h_split, h_split_out=h_split_out)

    # -- edge-range shards ------------------------------------------------------------------------------------
    def layer_raw(self, model, l: int, weights, h, h_split, plan, partial) -> None:
        """partial[v] = sum over THIS rank's edges into v of (h_u W_msg[r] + bias[r] + h_v W_self[r]) — no division, no tail."""
        W, W_self, bias = weights
        if plan.E == 0:
            partial.zero_()
            return
        _native.message_layer_fwd(h, plan, W, W_self, bias, plan.wlayout, None, None, 0.0, partial, h_split=h_split,
                                  flags=_native.GHF_FLAG_RAW_SUM)

    def scale_rows(self, sums: torch.Tensor, inv: torch.Tensor) -> torch.Tensor:
        return _native.rowscale(sums, inv)

    def tail_rows(self, model, l: int, agg, h, h_out, lo: int, hi: int) -> None:
        norm = model.layer_norms[l]
        _native.tail_fwd(agg, h, norm.weight.detach(), norm.bias.detach(), norm.eps, h_out, row0=lo, rows=hi - lo)


class ShardedHyperGNN:
    """Runs ``HyperGNN.forward`` across the ranks of a process group; every rank returns the full [N, d].

    mode: "dst" | "edges"; exchange: "allgather" | "pairs" | "sparse" (dst mode); balance: "rows" | "edges" (dst mode; "edges"
    implies a pairwise exchange, whose messages may differ in size).  Defaults from GHF_DIST_MODE / GHF_DIST_EXCHANGE /
    GHF_DIST_BALANCE / GHF_DIST_CHUNKS."""

    def __init__(self, model, group: Optional[dist.ProcessGroup] = None, ops=None, chunks: Optional[int] = None,
                 mode: Optional[str] = None, exchange: Optional[str] = None, balance: Optional[str] = None) -> None:
        self.model = model
        self.group = group
        self.ops = ops or NativeOps()
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.chunks = int(os.environ.get("GHF_DIST_CHUNKS", "4")) if chunks is None else chunks
        self.mode = mode or os.environ.get("GHF_DIST_MODE", "dst")
        self.balance = balance or os.environ.get("GHF_DIST_BALANCE", "rows")
        self.exchange = exchange or os.environ.get("GHF_DIST_EXCHANGE", "allgather")
        if self.mode not in ("dst", "edges") or self.balance not in ("rows", "edges") or self.exchange not in ("allgather", "pairs", "sparse"):
            raise ValueError(f"ShardedHyperGNN: mode={self.mode!r} balance={self.balance!r} exchange={self.exchange!r}")
        if self.balance == "edges" and self.exchange == "allgather":
            self.exchange = "pairs"
        self.backend = dist.get_backend(group)
        # what the backend offers is decided once, here — not by catching errors around a collective, which would hide real
        # RCCL failures and let one rank leave a collective the others are still in
        self._fused_gather = self.backend != "gloo"                      # all_gather_into_tensor / reduce_scatter_tensor
        self.profile = "full"                                            # "compute" / "exchange": bench.py's breakdown passes
        self.stats: Dict[str, float] = {}
        self._plans: Dict[Tuple, Tuple] = {}                             # key -> (plan, spec, 1/in-degree, inputs kept alive)
        self._plan = None
        self._spec: Optional[ShardSpec] = None
        self._inv = None
        self.last_range_flags = 0                                        # what the range guard saw in the last forward (OR over ranks)
        self._full_rows = False
        self._sparse: Optional[dict] = None
        self._comm_stream = None
        self._compute_streams: List = []

    # -- exchange ---------------------------------------------------------------------------------------
    def _gather_chunk(self, buf: torch.Tensor, spec: ShardSpec, c: int) -> None:
        """Make every rank's slot of chunk c of a row-indexed buffer visible on every rank."""
        if self.profile == "compute":
            return
        if self.exchange in ("pairs", "sparse") or not spec.uniform:
            return self._gather_pairs(buf, spec, c)
        lo, hi = spec.chunk_rows(c)
        if hi > buf.size(0):                                   # a buffer of exactly N rows: the chunk that reaches past N
            stage = torch.empty((hi - lo,) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
            a, b = spec.slot(c)
            if b > a:
                stage[a - lo: b - lo].copy_(buf[a:b])
            sub = ShardSpec(N=hi - lo, world=spec.world, rank=spec.rank, block_nodes=spec.block_nodes, chunks=1,
                            bounds=[g * spec.S for g in range(spec.world + 1)])
            self._gather_chunk(stage, sub, 0)
            if buf.size(0) > lo:
                buf[lo:].copy_(stage[: buf.size(0) - lo])
            return
        whole = buf[lo:hi]
        mine = buf[lo + spec.rank * spec.S: lo + (spec.rank + 1) * spec.S]
        self.stats["bytes_recv"] = self.stats.get("bytes_recv", 0.0) + (whole.numel() - mine.numel()) * whole.element_size()
        if self._fused_gather:
            dist.all_gather_into_tensor(whole, mine, group=self.group)
        elif buf.is_cuda:
            # gloo moves host memory only: bounce (this is how the multi-rank GPU test runs two ranks on one card)
            parts = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(self.world)]
            dist.all_gather(parts, mine.cpu(), group=self.group)
            whole.copy_(torch.cat(parts), non_blocking=False)
        else:
            parts = [buf[lo + g * spec.S: lo + (g + 1) * spec.S] for g in range(self.world)]
            dist.all_gather(parts, mine.clone(), group=self.group)

    def _gather_bufs(self, bufs: List[torch.Tensor], spec: ShardSpec, c: int) -> None:
        if self.profile == "compute":
            return
        if self.exchange == "sparse" and not self._full_rows:
            return self._gather_sparse(bufs, spec, c)
        for b in bufs:
            self._gather_chunk(b, spec, c)

    def _gather_pairs(self, buf: torch.Tensor, spec: ShardSpec, c: int) -> None:
        """Pairwise exchange of chunk c: my slot to every peer, every peer's slot from it (one batch of sends/receives).
        Only real rows travel (slots are clipped to N), and slots may differ in size."""
        nrows = buf.size(0)
        a, b = spec.slot(c)
        a, b = min(a, nrows), min(b, nrows)
        bounce = buf.is_cuda and self.backend == "gloo"
        mine = buf[a:b].cpu() if bounce else buf[a:b]
        ops, recvs = [], []
        for p in range(self.world):
            if p == self.rank:
                continue
            lo, hi = spec.slot(c, p)
            lo, hi = min(lo, nrows), min(hi, nrows)
            if b > a:
                ops.append(dist.P2POp(dist.isend, mine, p, group=self.group))
            if hi > lo:
                dst = torch.empty((hi - lo,) + tuple(buf.shape[1:]), dtype=buf.dtype) if bounce else buf[lo:hi]
                ops.append(dist.P2POp(dist.irecv, dst, p, group=self.group))
                recvs.append((lo, hi, dst))
                self.stats["bytes_recv"] = self.stats.get("bytes_recv", 0.0) + dst.numel() * dst.element_size()
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if bounce:
            for lo, hi, t in recvs:
                buf[lo:hi].copy_(t)

    def _gather_sparse(self, bufs: List[torch.Tensor], spec: ShardSpec, c: int) -> None:
        """Chunk c, needed rows only: to every peer the rows of my slot it asked for at plan time, from every peer the rows of
        its slot that my edges read, scattered to their places — ONE message per peer for all of `bufs` (the split form's
        rows and their scales travel together): ops.pack_rows -> pairwise send/recv -> ops.unpack_rows.  Rows nobody on this
        rank reads stay stale.  (ops without pack_rows — the CPU rehearsal's — go buffer by buffer through torch indexing.)"""
        send_idx, recv_idx = self._sparse["send"][c], self._sparse["recv"][c]
        if not hasattr(self.ops, "pack_rows") or not bufs[0].is_cuda:
            for buf in bufs:
                self._gather_sparse_torch(buf, send_idx, recv_idx)
            return
        bounce = self.backend == "gloo"
        ops, recvs, keep = [], [], []
        row_bytes = sum(b.size(1) * b.element_size() for b in bufs)
        for p in range(self.world):
            if p == self.rank:
                continue
            if send_idx[p].numel():
                packed = self.ops.pack_rows(bufs, send_idx[p])
                packed = packed.cpu() if bounce else packed
                keep.append(packed)
                ops.append(dist.P2POp(dist.isend, packed, p, group=self.group))
            if recv_idx[p].numel():
                t = torch.empty(recv_idx[p].numel(), row_bytes, dtype=torch.uint8, device="cpu" if bounce else bufs[0].device)
                ops.append(dist.P2POp(dist.irecv, t, p, group=self.group))
                recvs.append((recv_idx[p], t))
                self.stats["bytes_recv"] = self.stats.get("bytes_recv", 0.0) + t.numel()
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for idx, t in recvs:
            self.ops.unpack_rows(bufs, idx, t.to(bufs[0].device) if bounce else t)

    def _gather_sparse_torch(self, buf: torch.Tensor, send_idx, recv_idx) -> None:
        bounce = buf.is_cuda and self.backend == "gloo"
        ops, recvs, keep = [], [], []
        for p in range(self.world):
            if p == self.rank:
                continue
            if send_idx[p].numel():
                packed = buf.index_select(0, send_idx[p])
                packed = packed.cpu() if bounce else packed
                keep.append(packed)
                ops.append(dist.P2POp(dist.isend, packed, p, group=self.group))
            if recv_idx[p].numel():
                t = torch.empty((recv_idx[p].numel(),) + tuple(buf.shape[1:]), dtype=buf.dtype, device="cpu" if bounce else buf.device)
                ops.append(dist.P2POp(dist.irecv, t, p, group=self.group))
                recvs.append((recv_idx[p], t))
                self.stats["bytes_recv"] = self.stats.get("bytes_recv", 0.0) + t.numel() * t.element_size()
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for idx, t in recvs:
            buf.index_copy_(0, idx, t.to(buf.device) if bounce else t)

    def _plan_sparse(self, plan: GraphPlan, spec: ShardSpec, device) -> dict:
        """Who needs which rows: recv[c][p] = the rows of rank p's slot of chunk c that this rank's edges read (sorted row ids,
        on the device), send[c][q] = the rows of MY slot of chunk c that rank q reads.  One exchange of index lists per
        plan (host tensors through the group: counts by all_gather, lists pairwise)."""
        G, C, me = self.world, spec.chunks, self.rank
        # the lists travel through the group on the backend's own memory: host tensors over gloo, device tensors over RCCL
        comm = torch.device("cpu") if self.backend == "gloo" else torch.device(device)
        if plan.E > 0:
            need = torch.unique(self.ops.plan_sources(plan))
            slot = torch.bucketize(need, torch.tensor(spec.bounds[1:], dtype=torch.int64, device=need.device), right=True)
            need, slot = need.to(comm), slot.to(comm)
        else:
            need = slot = torch.zeros(0, dtype=torch.int64, device=comm)
        want = [[need[(slot == c * G + p)].contiguous() if p != me else need[:0] for p in range(G)] for c in range(C)]
        cnt = torch.tensor([[want[c][p].numel() for p in range(G)] for c in range(C)], dtype=torch.int64, device=comm)
        allc = [torch.zeros_like(cnt) for _ in range(G)]
        dist.all_gather(allc, cnt, group=self.group)                    # allc[q][c][p]: rows q needs from p in chunk c
        allc = [t.cpu() for t in allc]
        ops, bufs, keep = [], {}, []
        for p in range(G):
            if p == me:
                continue
            out = torch.cat([want[c][p] for c in range(C)]) if C else need[:0]
            if out.numel():
                keep.append(out)
                ops.append(dist.P2POp(dist.isend, out, p, group=self.group))
            n_in = int(allc[p][:, me].sum())
            if n_in:
                bufs[p] = torch.empty(n_in, dtype=torch.int64, device=comm)
                ops.append(dist.P2POp(dist.irecv, bufs[p], p, group=self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        empty = torch.zeros(0, dtype=torch.int64, device=device)
        send = [[empty] * G for _ in range(C)]
        for p, t in bufs.items():
            parts = torch.split(t, [int(allc[p][c, me]) for c in range(C)])
            for c in range(C):
                send[c][p] = parts[c].to(device)
        recv = [[want[c][p].to(device) for p in range(G)] for c in range(C)]
        cnt = cnt.cpu()
        owned = sum(hi - lo for lo, hi in spec.owned())
        return {"send": send, "recv": recv, "rows_needed": int(cnt.sum()), "rows_other": int(spec.N - owned)}

    def _run_chunked(self, bufs, spec: ShardSpec, compute_rows, full_rows: bool = False) -> None:
        """compute_rows(lo, hi) fills my rows of a chunk in every buffer of `bufs`; the chunk's exchange overlaps the next
        chunk's compute."""
        bufs = [bufs] if isinstance(bufs, torch.Tensor) else list(bufs)
        buf = bufs[0]
        self._full_rows = full_rows                            # (sparse exchange: this step's rows travel whole)
        on_gpu = buf.is_cuda
        skip_compute = self.profile == "exchange"
        if on_gpu:
            # The chunks of one step are independent, so their kernels go to alternating streams: a rank's chunk is a
            # fraction of a chip-filling launch (145 workgroups at 8 GPUs x 4 chunks), and in one stream every launch
            # would wait for the previous one's last workgroup (tools/dist_rank_cost.py, one rank of 8 at C3: 3.5 -> 2.7 ms per forward).
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=buf.device)
                self._compute_streams = [torch.cuda.Stream(device=buf.device) for _ in range(2)]
            main = torch.cuda.current_stream(buf.device)
            self._comm_stream.wait_stream(main)               # the previous users of `buf` are ordered before the gathers
            lanes = self._compute_streams if spec.chunks > 1 else [main]
            for s in lanes:
                if s is not main:
                    s.wait_stream(main)
        for c in range(spec.chunks):
            lo, hi = spec.slot(c)
            if on_gpu:
                lane = lanes[c % len(lanes)]
                try:                                          # (set_stream, not the context manager: that one costs two
                    torch.cuda.set_stream(lane)               #  slow current_stream() lookups per use)
                    if hi > lo and not skip_compute:
                        compute_rows(lo, hi)
                    ready = torch.cuda.Event()
                    ready.record(lane)
                    torch.cuda.set_stream(self._comm_stream)
                    self._comm_stream.wait_event(ready)
                    self._gather_bufs(bufs, spec, c)
                finally:
                    torch.cuda.set_stream(main)
            else:
                if hi > lo and not skip_compute:
                    compute_rows(lo, hi)
                self._gather_bufs(bufs, spec, c)
        if on_gpu:
            for s in lanes:
                if s is not main:
                    main.wait_stream(s)
            main.wait_stream(self._comm_stream)                # every row of `buf` is in place for the next layer

    # -- plan -------------------------------------------------------------------------------------------
    def plan_for(self, edge_index: torch.Tensor, edge_texts: Sequence[str], N: int, device, exact: bool = False) -> GraphPlan:
        """This rank's plan (and, in self._spec / self._inv, its shard geometry).  exact: the same shards planned for the exact
        fp32 kernels — what every rank reruns on when the range guard fired on any of them."""
        key = PlanCache.key(edge_index, edge_texts, N, self.model.hidden_dim, device,
                            extra=(self.world, self.rank, self.chunks, self.mode, self.balance, bool(exact)))
        hit = None if os.environ.get("GHF_PLAN_CACHE") == "0" else self._plans.get(key)
        if hit is None:
            d = self.model.hidden_dim
            unique, ids = relation_ids(edge_texts)
            E = edge_index.size(1)
            bn = self.ops.message_config(d, E, N, **({"exact": True} if exact else {}))[0]
            kw = {"exact": True} if exact else {}
            inv = None
            if self.mode == "edges":
                # contiguous ranges of the edge list; the rows are split evenly for the reduce-scatter / all-gather
                spec = shard_spec(N, 1, self.world, self.rank, 1)
                lo, hi = E * self.rank // self.world, E * (self.rank + 1) // self.world
                plan = self.ops.build_plan(edge_index, torch.from_numpy(ids), unique, N, d, device, edge_range=(lo, hi), **kw)
                deg = torch.bincount(edge_index[1].to(device), minlength=N).clamp_(min=1)
                inv = (1.0 / deg.to(torch.float32)).contiguous()
            else:
                block_edges = None
                if self.balance == "edges":
                    nb = -(-N // bn)
                    block_edges = torch.bincount(torch.div(edge_index[1], bn, rounding_mode="floor"), minlength=nb).cpu().tolist()
                spec = shard_spec(N, bn, self.world, self.rank, self.chunks, block_edges)
                plan = self.ops.build_plan(edge_index, torch.from_numpy(ids), unique, N, d, device, **spec.owner_arg(), **kw)
            sparse = None
            if self.mode == "dst" and self.world > 1 and (self.exchange == "sparse" or os.environ.get("GHF_DIST_ROW_STATS") == "1"):
                sparse = self._plan_sparse(plan, spec, device)
            hit = (plan, spec, inv, (edge_index, edge_texts), sparse)
            while len(self._plans) >= 4:
                self._plans.pop(next(iter(self._plans)))
            self._plans[key] = hit
        self._plan, self._spec, self._inv, self._sparse = hit[0], hit[1], hit[2], hit[4]
        return self._plan

    # -- forward ----------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, node_features: torch.Tensor, edge_index: torch.Tensor, edge_texts: List[str]) -> torch.Tensor:
        if edge_index.size(1) != len(edge_texts):
            raise ValueError(f"edge_index has {edge_index.size(1)} edges but edge_texts has {len(edge_texts)} entries")
        N, device = node_features.size(0), node_features.device
        plan = self.plan_for(edge_index, edge_texts, N, device)
        guard = isinstance(self.ops, NativeOps) and self.model._guarded(plan) and self.profile == "full"
        if guard:
            flag = _native.range_flag(device)
            flag.zero_()
        out = self._forward_on(node_features, plan)
        self.last_range_flags = 0
        if guard:
            # every rank must take the same decision: the guard bits are OR-ed across the ranks (one small collective: MAX
            # per bit — a MAX of the words would turn {1, 2} into 2)
            word = flag.cpu() if self.backend == "gloo" else flag
            bits = torch.stack([(word >> i) & 1 for i in range(3)]).flatten()
            dist.all_reduce(bits, op=dist.ReduceOp.MAX, group=self.group)
            word = sum(int(b) << i for i, b in enumerate(bits.tolist()))
            self.last_range_flags = word
            if word:
                # Some row of h or some relation's weights spans more dynamic range than two fp16 pieces hold (include/ghf.h:
                # ghf_set_range_flag).  The single-GPU forward reruns on the exact fp32 kernels (HyperGNN._forward_exact), and
                # so does this one — on every rank, since the reduced word is the same everywhere: the same shards planned for
                # the exact kernels (reference: plain fp32 bmm, hypergnn.py:202,228).
                out = self._forward_on(node_features, self.plan_for(edge_index, edge_texts, N, device, exact=True))
        return out

    def _forward_on(self, node_features: torch.Tensor, plan: GraphPlan) -> torch.Tensor:
        model, spec = self.model, self._spec
        N, device, d = node_features.size(0), node_features.device, model.hidden_dim
        self.stats = {"bytes_recv": 0.0}
        # fresh buffers per call (the result is a view of one of them); pad rows are exchanged but never read
        h = torch.empty(spec.padded_rows, d, dtype=torch.float32, device=device)
        h_next = torch.empty_like(h)
        text_embs = self.ops.text_embs(model, plan.unique_texts, device)
        if self.mode == "edges":
            return self._forward_edges(node_features, plan, spec, text_embs, h, h_next)
        if self.ops.exchanges_split(plan):
            return self._forward_split(node_features, plan, spec, text_embs, h, h_next)
        return self._forward_rows(node_features, plan, spec, text_embs, h, h_next)

    def _forward_rows(self, node_features, plan, spec, text_embs, h, h_next) -> torch.Tensor:
        model, N, device = self.model, node_features.size(0), node_features.device
        all_w, ready = self.ops.all_weights(model, text_embs, plan)
        self.ops.input_proj(model, node_features, h[:N], None, plan)
        for l in range(model.num_layers):
            if ready is not None and ready[l] is not None:
                torch.cuda.current_stream(device).wait_event(ready[l])
            weights = all_w[l]
            src, dst = h, h_next
            self.ops.layer_begin(model, l, weights, src[:N], plan)
            src_split = self.ops.split_rows(plan, src[:N])          # all rows are in place after the previous exchange
            self._run_chunked(dst, spec, lambda lo, hi: self.ops.layer_rows(model, l, weights, src[:N], src_split, plan,
                                                                            dst[:N], lo, hi),
                              full_rows=l == model.num_layers - 1)
            h, h_next = h_next, h
        return h[:N]

    def _forward_split(self, node_features, plan, spec, text_embs, h, h_next) -> torch.Tensor:
        """Layers whose kernel gathers pre-split rows: a rank keeps fp32 h for its own rows only (the residual input of
        the next layer) and the ranks exchange the split rows their fused tails wrote; fp32 rows travel once, at the end."""
        model, ops = self.model, self.ops
        N, d, device = node_features.size(0), model.hidden_dim, node_features.device
        hs, hs_next = ops.alloc_split(plan, N, d, device), ops.alloc_split(plan, N, d, device)

        te_done = None
        if node_features.is_cuda:
            te_done = torch.cuda.Event()
            te_done.record(torch.cuda.current_stream(device))
        ops.input_proj(model, node_features, h[:N], hs, plan)
        # enqueued after the projection so that its kernel is not queued behind the generators' on a shared hardware
        # queue; the side stream itself only waits for the text embeddings
        all_w, ready = ops.all_weights(model, text_embs, plan, after=te_done)
        last = model.num_layers - 1
        for l in range(model.num_layers):
            if ready is not None and ready[l] is not None:
                torch.cuda.current_stream(device).wait_event(ready[l])
            weights = all_w[l]
            src, dst, src_split = h, h_next, hs
            out_split = None if l == last else hs_next
            self._run_chunked(dst if l == last else ops.split_parts(plan, hs_next, N, d), spec,
                              lambda lo, hi: ops.layer_rows(model, l, weights, src[:N], src_split, plan, dst[:N], lo, hi,
                                                            h_split_out=out_split), full_rows=l == last)
            h, h_next = h_next, h
            hs, hs_next = hs_next, hs
        return h[:N]

    def _forward_edges(self, node_features, plan, spec, text_embs, h, h_next) -> torch.Tensor:
        """Edge-range shards (BASELINE config 4 as written): per layer raw partial sums over my edges -> reduction across
        ranks -> my rows: mean, tail -> all-gather of the new rows."""
        model, ops = self.model, self.ops
        N, d, device = node_features.size(0), model.hidden_dim, node_features.device
        G, S = self.world, spec.S
        all_w, ready = ops.all_weights(model, text_embs, plan)
        ops.input_proj(model, node_features, h[:N], None, plan)
        partial = torch.empty(spec.padded_rows, d, dtype=torch.float32, device=device)
        if spec.padded_rows > N:
            partial[N:].zero_()
        lo, hi = spec.slot(0)
        for l in range(model.num_layers):
            if ready is not None and ready[l] is not None:
                torch.cuda.current_stream(device).wait_event(ready[l])
            weights = all_w[l]
            src_split = ops.split_rows(plan, h[:N])
            if self.profile != "exchange":
                ops.layer_raw(model, l, weights, h[:N], src_split, plan, partial[:N])
            agg = h_next                                             # (scratch until the tail overwrites my rows of it)
            if self.profile != "compute":
                self.stats["bytes_recv"] += (G - 1) * S * d * 4
                if self._fused_gather:
                    dist.reduce_scatter_tensor(agg[self.rank * S:(self.rank + 1) * S], partial, group=self.group)
                else:
                    red = partial.cpu() if partial.is_cuda else partial
                    dist.all_reduce(red, group=self.group)           # backends without reduce-scatter (gloo rehearsals)
                    agg[self.rank * S:(self.rank + 1) * S].copy_(red[self.rank * S:(self.rank + 1) * S])
            if hi > lo and self.profile != "exchange":
                agg[lo:hi] = ops.scale_rows(agg[lo:hi], self._inv[lo:hi])
                ops.tail_rows(model, l, agg[:N], h[:N], partial[:N], lo, hi)       # (partial: free after the reduction)
                h_next[lo:hi].copy_(partial[lo:hi])
            self._gather_chunk(h_next, spec, 0)
            h, h_next = h_next, h
        return h[:N]

    __call__ = forward
