"""Multi-GPU forward: one process per GPU, destination-range sharding, RCCL all-gather.

The path shards by DESTINATION node: rank g owns rows [g*S, (g+1)*S) (S a multiple of
the kernel's destination-block size), i.e. every in-edge of its nodes.  Each edge is
owned by exactly one rank, the per-destination sums never cross ranks, and the fused
tail stays local; the one real exchange step per layer is making the new h visible
everywhere: an in-place all-gather of [S, d] fp32 shards over xGMI (half the bytes of
the all-reduce an edge-range partition would need — SURVEY.md §8e).  The input
projection is sharded the same way.  Weight generation (0.5 GFLOP) is recomputed on
every rank instead of broadcast.

The compute steps are taken from an `ops` object so that the sharding/exchange logic
can be exercised on CPU with gloo (tests/test_dist_gloo.py injects the oracle there);
the product default, NativeOps, calls the HIP library and nothing else.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _native
from .plan import GraphPlan, build_plan, relation_ids


@dataclass
class ShardSpec:
    N: int
    world: int
    rank: int
    block_nodes: int
    S: int                      # rows per rank (multiple of block_nodes)

    @property
    def padded_rows(self) -> int:
        return self.S * self.world

    @property
    def lo(self) -> int:
        return min(self.N, self.rank * self.S)

    @property
    def hi(self) -> int:
        return min(self.N, (self.rank + 1) * self.S)


def shard_spec(N: int, block_nodes: int, world: int, rank: int) -> ShardSpec:
    nb = -(-N // block_nodes)
    S = -(-nb // world) * block_nodes
    return ShardSpec(N=N, world=world, rank=rank, block_nodes=block_nodes, S=S)


class NativeOps:
    """The product compute steps: C-ABI calls only."""

    def message_config(self, d: int):
        return _native.message_config(d)

    def build_plan(self, edge_index, rel_ids, unique, N, d, device, row_range) -> GraphPlan:
        return build_plan(edge_index, rel_ids, unique, N, d, device, row_range=row_range)

    def text_embs(self, model, unique: Sequence[str], device) -> torch.Tensor:
        return model.text_encoder(unique, device)

    def input_proj(self, model, x_rows: torch.Tensor, out_rows: torch.Tensor) -> None:
        _native.input_proj_fwd(x_rows, model.input_proj.weight.detach(), model.input_proj.bias.detach(), out=out_rows)

    def layer(self, model, l: int, text_embs, h, plan, h_out, lo: int, hi: int) -> None:
        gen, norm = model.weight_generators[l], model.layer_norms[l]
        W, W_self, bias = gen.generate(text_embs, plan.wlayout)
        _native.message_layer_fwd(h, plan, W, W_self, bias, plan.wlayout, norm.weight.detach(), norm.bias.detach(),
                                  norm.eps, h_out, row0=lo, rows=hi - lo)


class ShardedHyperGNN:
    """Runs ``HyperGNN.forward`` across the ranks of a process group; every rank returns the full [N, d]."""

    def __init__(self, model, group: Optional[dist.ProcessGroup] = None, ops=None) -> None:
        self.model = model
        self.group = group
        self.ops = ops or NativeOps()
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._plan_key = None
        self._plan = None
        self._spec: Optional[ShardSpec] = None

    def _exchange(self, buf: torch.Tensor, spec: ShardSpec) -> None:
        """In-place all-gather: rank g contributes rows [g*S, (g+1)*S) of the padded buffer."""
        if self.world == 1:
            return
        mine = buf[spec.rank * spec.S:(spec.rank + 1) * spec.S]
        try:
            dist.all_gather_into_tensor(buf, mine, group=self.group)
        except (RuntimeError, NotImplementedError):          # backends without the fused form
            parts = [buf[g * spec.S:(g + 1) * spec.S] for g in range(self.world)]
            dist.all_gather(parts, mine.clone(), group=self.group)

    def plan_for(self, edge_index: torch.Tensor, edge_texts: Sequence[str], N: int, device) -> GraphPlan:
        key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, id(edge_texts), len(edge_texts), N)
        if key != self._plan_key:
            bn = self.ops.message_config(self.model.hidden_dim)[0]
            spec = shard_spec(N, bn, self.world, self.rank)
            unique, ids = relation_ids(edge_texts)
            self._plan = self.ops.build_plan(edge_index, torch.from_numpy(ids), unique, N, self.model.hidden_dim,
                                             device, (spec.lo, spec.hi))
            self._spec, self._plan_key, self._keep = spec, key, (edge_index, edge_texts)
        return self._plan

    @torch.no_grad()
    def forward(self, node_features: torch.Tensor, edge_index: torch.Tensor, edge_texts: List[str]) -> torch.Tensor:
        model = self.model
        if edge_index.size(1) != len(edge_texts):
            raise ValueError(f"edge_index has {edge_index.size(1)} edges but edge_texts has {len(edge_texts)} entries")
        N, device = node_features.size(0), node_features.device
        plan = self.plan_for(edge_index, edge_texts, N, device)
        spec = self._spec
        d = model.hidden_dim
        h = torch.zeros(spec.padded_rows, d, dtype=torch.float32, device=device)
        h_next = torch.zeros_like(h)
        lo, hi = spec.lo, spec.hi
        text_embs = self.ops.text_embs(model, plan.unique_texts, device)
        if hi > lo:
            self.ops.input_proj(model, node_features[lo:hi], h[lo:hi])
        self._exchange(h, spec)
        for l in range(model.num_layers):
            if hi > lo:
                self.ops.layer(model, l, text_embs, h[:N], plan, h_next[:N], lo, hi)
            self._exchange(h_next, spec)
            h, h_next = h_next, h
        return h[:N]

    __call__ = forward
