"""Multi-GPU forward: one process per GPU, destination sharding, RCCL all-gather overlapped with compute.

The path shards by DESTINATION node: a rank owns every in-edge of its rows, so each edge is
owned by exactly one rank, the per-destination sums never cross ranks and the fused tail
stays local.  The one real exchange step per layer is making the new h visible everywhere:
an all-gather of rows over xGMI (half the bytes of the all-reduce an edge-range
partition would need — SURVEY.md §8e).  The input projection is not exchanged: every rank computes it for all rows.  What travels is what the next layer's kernel gathers:
for the default d = 128 kernel the rows already cut into their two fp16 pieces plus the row
scales (written by the layer kernel's fused tail, 4 d + 4 bytes per row, the same volume as
fp32), so no rank ever re-splits the full h; fp32 rows travel only after the last layer, and
for the kernels that gather fp32 rows themselves.

Ownership is block-cyclic so that the exchange overlaps the compute: the (padded) rows are cut
into C chunks of G*S rows, and inside chunk c rank g owns rows [(c*G+g)*S, (c*G+g+1)*S)
(S a multiple of the kernel's destination-block size).  A layer then runs as
    for c in chunks:  launch the message kernel on my rows of chunk c          (compute stream)
                      all-gather chunk c in place: a contiguous [G*S, d] slice   (comm stream)
so the gather of chunk c travels while chunk c+1 computes, and only the last chunk's gather
is exposed.  Weight generation (0.5 GFLOP) and the input projection (33 GFLOP) are recomputed on every rank
instead of exchanged.

The compute steps come from an `ops` object so that the sharding / exchange logic can be
exercised on CPU with gloo (tests/test_dist_gloo.py injects the oracle there); the product
default, NativeOps, calls the HIP library and nothing else.
"""

from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _native
from .plan import GraphPlan, build_plan, build_rs, relation_ids


@dataclass
class ShardSpec:
    N: int
    world: int
    rank: int
    block_nodes: int
    chunks: int                 # C
    S: int                      # rows per (rank, chunk) slot; multiple of block_nodes

    @property
    def padded_rows(self) -> int:
        return self.S * self.world * self.chunks

    def slot(self, c: int, g: Optional[int] = None) -> Tuple[int, int]:
        """Row range [lo, hi) (clipped to N) that rank g owns in chunk c."""
        g = self.rank if g is None else g
        lo = (c * self.world + g) * self.S
        return min(self.N, lo), min(self.N, lo + self.S)

    def chunk_rows(self, c: int) -> Tuple[int, int]:
        """Padded row range of chunk c (all ranks' slots): what one all-gather fills."""
        return c * self.world * self.S, (c + 1) * self.world * self.S

    def owned(self) -> List[Tuple[int, int]]:
        return [r for r in (self.slot(c) for c in range(self.chunks)) if r[1] > r[0]]


def shard_spec(N: int, block_nodes: int, world: int, rank: int, chunks: int = 1) -> ShardSpec:
    nb = -(-N // block_nodes)
    chunks = max(1, min(chunks, -(-nb // world)))          # no more chunks than blocks per rank
    S = -(-nb // (world * chunks)) * block_nodes
    return ShardSpec(N=N, world=world, rank=rank, block_nodes=block_nodes, chunks=chunks, S=S)


class NativeOps:
    """The product compute steps: C-ABI calls only."""

    def message_config(self, d: int):
        return _native.message_config(d)

    def build_plan(self, edge_index, rel_ids, unique, N, d, device, owner) -> GraphPlan:
        return build_plan(edge_index, rel_ids, unique, N, d, device, owner=owner,
                          force_generic=_native.prefer_rs(d, len(unique)))

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

    def layer_begin(self, model, l: int, weights, h, plan) -> None:
        """Once per layer, before its chunks.  Wide rows (csrc/message_rs.hip): pass 1 over all of this rank's edges —
        it runs in relation order, not by destination chunk; the chunks then only do pass 2 on their rows."""
        if plan.block_nodes == 1 and _native.rs_supported(h.size(1)) and plan.E > 0:
            if plan.rs is None:
                plan.rs = build_rs(plan)
            W_msg, W_self, bias = weights
            _native.edge_transform_fwd(h, plan.rs, W_msg, W_self, bias, plan.rs.scratch(plan.E, h.size(1), h.device))

    def layer_rows(self, model, l: int, weights, h, h_split, plan, h_out, lo: int, hi: int, h_split_out=None) -> None:
        norm = model.layer_norms[l]
        W, W_self, bias = weights
        if plan.block_nodes == 1 and _native.rs_supported(h.size(1)):
            if plan.rs is None:
                plan.rs = build_rs(plan)
            _native.segment_tail_fwd(plan.rs.scratch(plan.E, h.size(1), h.device), plan.rs, h, norm.weight.detach(),
                                     norm.bias.detach(), norm.eps, h_out, row0=lo, rows=hi - lo)
            return
        _native.message_layer_fwd(h, plan, W, W_self, bias, plan.wlayout, norm.weight.detach(), norm.bias.detach(),
                                  norm.eps, h_out, row0=lo, rows=hi - lo, h_split=h_split, h_split_out=h_split_out)


class ShardedHyperGNN:
    """Runs ``HyperGNN.forward`` across the ranks of a process group; every rank returns the full [N, d]."""

    def __init__(self, model, group: Optional[dist.ProcessGroup] = None, ops=None, chunks: Optional[int] = None) -> None:
        self.model = model
        self.group = group
        self.ops = ops or NativeOps()
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.chunks = int(os.environ.get("GHF_DIST_CHUNKS", "4")) if chunks is None else chunks
        self._plan_key = None
        self._plan = None
        self._spec: Optional[ShardSpec] = None
        self._comm_stream = None
        self._compute_streams: List = []

    # -- exchange ---------------------------------------------------------------------------------------
    def _gather_chunk(self, buf: torch.Tensor, spec: ShardSpec, c: int) -> None:
        """All-gather of chunk c of a row-indexed buffer: rank g contributes its slot of the contiguous chunk slice.
        In place when the buffer has the padded rows; a buffer of exactly N rows takes the chunk that reaches past N
        through a staging copy."""
        lo, hi = spec.chunk_rows(c)
        if hi > buf.size(0):
            stage = torch.empty((hi - lo,) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
            a, b = spec.slot(c)
            if b > a:
                stage[a - lo: b - lo].copy_(buf[a:b])
            self._gather_chunk(stage, ShardSpec(N=hi - lo, world=spec.world, rank=spec.rank, block_nodes=spec.block_nodes,
                                                chunks=1, S=spec.S), 0)
            if buf.size(0) > lo:
                buf[lo:].copy_(stage[: buf.size(0) - lo])
            return
        whole = buf[lo:hi]
        mine = buf[lo + spec.rank * spec.S: lo + (spec.rank + 1) * spec.S]
        if buf.is_cuda and dist.get_backend(self.group) == "gloo":
            # gloo moves host memory only: bounce (this is how the multi-rank GPU test runs two ranks on one card)
            parts = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(self.world)]
            dist.all_gather(parts, mine.cpu(), group=self.group)
            whole.copy_(torch.cat(parts), non_blocking=False)
            return
        try:
            dist.all_gather_into_tensor(whole, mine, group=self.group)
        except (RuntimeError, NotImplementedError):          # backends without the fused form
            parts = [buf[lo + g * spec.S: lo + (g + 1) * spec.S] for g in range(self.world)]
            dist.all_gather(parts, mine.clone(), group=self.group)

    def _run_chunked(self, bufs, spec: ShardSpec, compute_rows) -> None:
        """compute_rows(lo, hi) fills my rows of a chunk in every buffer of `bufs`; the chunk's gathers overlap the next
        chunk's compute."""
        bufs = [bufs] if isinstance(bufs, torch.Tensor) else list(bufs)
        buf = bufs[0]
        on_gpu = buf.is_cuda
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
                    if hi > lo:
                        compute_rows(lo, hi)
                    ready = torch.cuda.Event()
                    ready.record(lane)
                    torch.cuda.set_stream(self._comm_stream)
                    self._comm_stream.wait_event(ready)
                    for b in bufs:
                        self._gather_chunk(b, spec, c)
                finally:
                    torch.cuda.set_stream(main)
            else:
                if hi > lo:
                    compute_rows(lo, hi)
                for b in bufs:
                    self._gather_chunk(b, spec, c)
        if on_gpu:
            for s in lanes:
                if s is not main:
                    main.wait_stream(s)
            main.wait_stream(self._comm_stream)                # every row of `buf` is in place for the next layer

    # -- plan -------------------------------------------------------------------------------------------
    def plan_for(self, edge_index: torch.Tensor, edge_texts: Sequence[str], N: int, device) -> GraphPlan:
        key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, id(edge_texts), len(edge_texts), N)
        if key != self._plan_key:
            bn = self.ops.message_config(self.model.hidden_dim)[0]
            spec = shard_spec(N, bn, self.world, self.rank, self.chunks)
            unique, ids = relation_ids(edge_texts)
            self._plan = self.ops.build_plan(edge_index, torch.from_numpy(ids), unique, N, self.model.hidden_dim,
                                             device, (spec.S, self.world, self.rank))
            self._spec, self._plan_key, self._keep = spec, key, (edge_index, edge_texts)
        return self._plan

    # -- forward ----------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, node_features: torch.Tensor, edge_index: torch.Tensor, edge_texts: List[str]) -> torch.Tensor:
        model = self.model
        if edge_index.size(1) != len(edge_texts):
            raise ValueError(f"edge_index has {edge_index.size(1)} edges but edge_texts has {len(edge_texts)} entries")
        N, device = node_features.size(0), node_features.device
        plan = self.plan_for(edge_index, edge_texts, N, device)
        spec = self._spec
        d = model.hidden_dim
        # fresh buffers per call (the result is a view of one of them); pad rows are exchanged but never read
        h = torch.empty(spec.padded_rows, d, dtype=torch.float32, device=device)
        h_next = torch.empty_like(h)
        text_embs = self.ops.text_embs(model, plan.unique_texts, device)
        if self.ops.exchanges_split(plan):
            return self._forward_split(node_features, plan, spec, text_embs, h, h_next)
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
                                                                            dst[:N], lo, hi))
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
                                                            h_split_out=out_split))
            h, h_next = h_next, h
            hs, hs_next = hs_next, hs
        return h[:N]

    __call__ = forward
