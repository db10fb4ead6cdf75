"""HyperGNN — host mirror of the reference model, computing on MI355X.

Same constructor, attributes, ``state_dict`` keys, ``forward(node_features,
edge_index, edge_texts)`` signature and ``ValueError`` behaviour as
``graph_hypernetwork_forge/models/hypergnn.py:88-322`` of the reference.  The
forward is a sequence of C-ABI calls into ``libghf_hip.so``:

    plan (cached)            ghf_plan_build            replaces hypergnn.py:264-268 + edge order
    h0 = relu(x W_in^T + b)  ghf_input_proj_fwd        replaces :261
    per layer:
      weights per relation   ghf_weightgen_fwd         replaces :278 (weight_generator.py:137-141)
      messages+mean+self+tail ghf_message_layer_fwd    replaces :281-296

The per-edge weight gather of the reference (:281-283, O(E d^2) memory) does
not exist here: kernels index W[r] in place.
"""

from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import os

import numpy as np
import torch
import torch.nn as nn

from .. import _native
from ..plan import GraphPlan, PlanCache, build_plan, build_rs, exact_plan, relation_ids
from ..plan import check_pool
from .weight_generator import WeightGenerator, check_dropout, draw_mask, require_inference, wants_grad


class TextEncoder(nn.Module):
    """Relation string -> ``[text_dim]``: mean of character embeddings, Linear, Tanh.

    Mirrors reference hypergnn.py:39-81 (ids = min(ord(c), 127), '' -> [0]).
    ``forward`` encodes all strings with one ``ghf_text_encode_fwd`` launch; the
    padded id matrix of a list of strings is built once and kept on the device
    (keyed on the tuple of strings), so a warm forward has no host work here.
    """

    ASCII_VOCAB = 128

    def __init__(self, text_dim: int, char_emb_dim: int = 32) -> None:
        super().__init__()
        self.text_dim = text_dim
        self.char_emb = nn.Embedding(self.ASCII_VOCAB, char_emb_dim)
        self.proj = nn.Sequential(nn.Linear(char_emb_dim, text_dim), nn.Tanh())
        self._tokens: Dict[Tuple, Tuple[torch.Tensor, torch.Tensor]] = {}      # padded id matrices, on the device

    def _codes(self, text: str) -> List[int]:
        codes = [min(ord(c), self.ASCII_VOCAB - 1) for c in text]
        return codes or [0]

    def _tokenize(self, text: str, device: torch.device) -> torch.Tensor:
        return torch.tensor(self._codes(text), dtype=torch.long, device=device)

    def _token_matrix(self, texts: Sequence[str], device: torch.device):
        key = (tuple(texts), str(device))
        hit = self._tokens.get(key)
        if hit is None:
            codes = [self._codes(t) for t in texts]
            lens = np.fromiter((len(c) for c in codes), dtype=np.int32, count=len(codes))
            ids = np.zeros((len(codes), int(lens.max())), dtype=np.int32)
            for i, c in enumerate(codes):
                ids[i, :len(c)] = c
            hit = (torch.from_numpy(ids).to(device), torch.from_numpy(lens).to(device))
            if len(self._tokens) >= 8:
                self._tokens.pop(next(iter(self._tokens)))
            self._tokens[key] = hit
        return hit

    def encode_one(self, text: str, device: torch.device) -> torch.Tensor:
        """One string -> ``[text_dim]`` (reference hypergnn.py:73-77)."""
        return self.forward([text], device)[0]

    def forward(self, texts: Sequence[str], device: torch.device) -> torch.Tensor:
        grad = wants_grad(self, self.char_emb.weight)
        ids, lens = self._token_matrix(texts, torch.device(device))
        lin = self.proj[0]
        if grad:
            from ..autograd import TextEncoderFn
            return TextEncoderFn.apply(self.char_emb.weight, lin.weight, lin.bias, ids, lens)
        return _native.text_encode_fwd(ids, lens, self.char_emb.weight.detach(), lin.weight.detach(), lin.bias.detach())


class GraphedForward:
    """The warm inference forward of one (model, graph plan, feature shape) captured into a HIP graph.

    A forward is ~6 + 4 L kernel launches; on small graphs (BASELINE configs 1 and 2) their host cost exceeds the
    device time.  Replaying the captured graph costs one launch.  Parameters are read at replay time (in-place updates
    are seen); the plan, the feature shape and the relation strings are frozen.  ``replay(node_features)`` copies new
    features into the captured input buffer; the returned tensor is overwritten by the next replay."""

    def __init__(self, model: "HyperGNN", node_features: torch.Tensor, plan: GraphPlan) -> None:
        self.input = node_features.detach().float().clone()
        dev = self.input.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():            # warm-up off the capture: lazy scratch, LDS limits
            for _ in range(2):
                model.forward_planned(self.input, plan)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.output = model.forward_planned(self.input, plan)
        # The captured launches hold raw device pointers into the plan's arrays (and its scratch), the relation strings'
        # token matrices and the parameters; the plan cache and the token cache are small LRUs, so this object keeps its
        # own references — evicting the plan elsewhere must not free memory a replay still reads.
        self._keep = (model, plan, plan.rs, plan._partial, None if plan.rs is None else (plan.rs._Y, plan.rs._P),
                      model.text_encoder._token_matrix(plan.unique_texts, dev))

    def replay(self, node_features: Optional[torch.Tensor] = None) -> torch.Tensor:
        if node_features is not None:
            if node_features.shape != self.input.shape:
                raise ValueError(f"captured for features {tuple(self.input.shape)}, got {tuple(node_features.shape)}")
            self.input.copy_(node_features)
        model, plan = self._keep[0], self._keep[1]
        guard = model._guarded(plan)
        if guard:
            flag = _native.range_flag(self.input.device)
            flag.zero_()
        self.graph.replay()
        if guard and int(flag.item()):                      # (one 4-byte read per replay; GHF_RANGE_GUARD=0 skips it)
            with torch.no_grad():
                self.output.copy_(model._forward_exact(self.input, plan, int(flag.item())))
        return self.output

    __call__ = replay


class HyperGNN(nn.Module):
    """Hypernetwork-conditioned GNN (reference hypergnn.py:88-154), forward on HIP kernels."""

    SIDE_STREAM_MIN_EDGES = 4_000_000     # below this a forward is too short for cross-stream overlap to pay

    def __init__(self, text_dim: int, node_feat_dim: int, hidden_dim: int, num_layers: int = 2,
                 dropout: float = 0.0, char_emb_dim: int = 32) -> None:
        super().__init__()
        if num_layers < 1:                                            # reference :123-124
            raise ValueError("num_layers must be at least 1")
        check_dropout(dropout)
        self.text_dim, self.node_feat_dim, self.hidden_dim = text_dim, node_feat_dim, hidden_dim
        self.num_layers, self.dropout = num_layers, dropout
        self.text_encoder = TextEncoder(text_dim=text_dim, char_emb_dim=char_emb_dim)
        self.input_proj = nn.Linear(node_feat_dim, hidden_dim)
        self.weight_generators = nn.ModuleList([
            WeightGenerator(text_dim=text_dim, d_in=hidden_dim, d_out=hidden_dim,
                            hidden_dim=max(64, text_dim * 2), num_hidden=2, dropout=dropout)
            for _ in range(num_layers)])
        self.layer_norms = nn.ModuleList([nn.LayerNorm(hidden_dim) for _ in range(num_layers)])
        self._plans = PlanCache()
        self._wg_stream = None
        self.last_range_flags = 0        # what the range guard saw in the last forward (include/ghf.h: ghf_set_range_flag)

    # -- plan ------------------------------------------------------------------------------
    def plan_for(self, edge_index: torch.Tensor, edge_texts: Sequence[str], num_nodes: int,
                 device: torch.device, training: bool = False) -> GraphPlan:
        """Cached graph plan for these inputs (cold: O(E) host work + one device sort).  Inference plans of graphs with
        many relations are CSR plans for the relation-stationary layer (_native.prefer_rs); plans that will record
        gradients keep the destination-block geometry the backward kernels run on."""
        return self._plan_lookup(edge_index, edge_texts, num_nodes, device, training)[0]

    def _plan_lookup(self, edge_index: torch.Tensor, edge_texts: Sequence[str], num_nodes: int, device: torch.device,
                     training: bool = False, defer_check: bool = False):
        """(plan, check, settle): the cached plan, or a fresh one (settle: None, or — with `defer_check` — a callable the caller
        runs after its launches are enqueued: it waits for the new cache entry's checksums, taken on the cache's threads
        meanwhile).  A hit on a relation list too long for the key's fingerprint to
        cover whole (plan.FULL_FINGERPRINT_MAX) is confirmed against the snapshot of the list taken when the plan was built
        (plan.same_relations: ~4 ms at 10 M entries) — here, before returning, or, with `defer_check`, by the caller:
        `check` is then a callable () -> bool to run once the forward is enqueued (the host is idle while the GPU works);
        False means the list was edited in place, the result must be dropped and the lookup repeated (it will miss)."""
        key = PlanCache.key(edge_index, edge_texts, num_nodes, self.hidden_dim, device) + (bool(training),)
        plan = self._plans.get(key)
        if plan is not None:
            check = self._plans.verifier(key, edge_texts)
            if check is None:
                return plan, None, None
            if defer_check:
                return plan, check, None
            if check():
                return plan, None, None
        unique, ids, objects = relation_ids(edge_texts, want_objects=True)
        wide = not training and _native.prefer_rs(self.hidden_dim, len(unique))
        plan = build_plan(edge_index, torch.from_numpy(ids), unique, num_nodes, self.hidden_dim, device, force_generic=wide)
        # (defer_check: the caller collects the entry's checksums — taken on the cache's threads beside its launches)
        taking = self._plans.put(key, plan, edge_index, edge_texts, objects=objects, background=defer_check)
        return plan, None, (None if taking is None else taking.result)

    def graphed(self, node_features: torch.Tensor, edge_index: torch.Tensor, edge_texts: List[str]) -> GraphedForward:
        """Capture ``forward`` for these inputs into a HIP graph (inference only); see GraphedForward."""
        if edge_index.size(1) != len(edge_texts):
            raise ValueError(f"edge_index has {edge_index.size(1)} edges but edge_texts has {len(edge_texts)} entries")
        if node_features.dim() != 2 or node_features.size(1) != self.node_feat_dim:
            raise ValueError(f"node_features must be [N, {self.node_feat_dim}], got {tuple(node_features.shape)}")
        if not node_features.is_cuda:
            raise RuntimeError("HyperGNN computes on an MI355X HIP device only; there is no CPU path to capture")
        plan = self.plan_for(edge_index, edge_texts, node_features.size(0), node_features.device)
        return GraphedForward(self, node_features, plan)

    def clear_plan_cache(self) -> None:
        self._plans.clear()

    def forward_ids(self, node_features: torch.Tensor, edge_index: torch.Tensor, edge_rel_ids: torch.Tensor,
                    relation_texts: Sequence[str]) -> torch.Tensor:
        """``forward`` for callers that already hold relation ids: ``edge_texts[e] == relation_texts[edge_rel_ids[e]]``.

        The reference's call form hands over one Python string per edge; mapping those to ids is pure host work
        (1.5 s at 10 M edges, reference hypergnn.py:264-268).  This overload skips it: the ids
        may live on the device, nothing O(E) runs on the host, and the plan is cached on the two tensors' identity."""
        if edge_rel_ids.dim() != 1 or edge_index.size(1) != edge_rel_ids.numel():
            raise ValueError(f"edge_index has {edge_index.size(1)} edges but edge_rel_ids has {edge_rel_ids.numel()} entries")
        if node_features.dim() != 2 or node_features.size(1) != self.node_feat_dim:
            raise ValueError(f"node_features must be [N, {self.node_feat_dim}], got {tuple(node_features.shape)}")
        grad = wants_grad(self, node_features) or self._dropping()
        device, N = node_features.device, node_features.size(0)
        texts = list(relation_texts)
        key = ("ids", edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, str(edge_index.device),
               edge_rel_ids.data_ptr(), edge_rel_ids._version, str(edge_rel_ids.device), tuple(texts), N, self.hidden_dim,
               str(device), bool(grad))
        plan = self._plans.get(key)
        if plan is None:
            wide = not grad and _native.prefer_rs(self.hidden_dim, len(texts))
            plan = build_plan(edge_index, edge_rel_ids, texts, N, self.hidden_dim, device, force_generic=wide)   # ids out of range: IndexError
            self._plans.put(key, plan, edge_index, (edge_rel_ids, texts))
        if grad:
            return self._forward_recorded(node_features, plan, edge_index)
        return self.forward_planned(node_features, plan)

    # -- forward (reference :236-298) -----------------------------------------------------
    def forward(self, node_features: torch.Tensor, edge_index: torch.Tensor, edge_texts: List[str]) -> torch.Tensor:
        if edge_index.size(1) != len(edge_texts):                     # reference :252-256
            raise ValueError(f"edge_index has {edge_index.size(1)} edges but "
                             f"edge_texts has {len(edge_texts)} entries")
        if node_features.dim() != 2 or node_features.size(1) != self.node_feat_dim:
            raise ValueError(f"node_features must be [N, {self.node_feat_dim}], got {tuple(node_features.shape)}")
        grad = wants_grad(self, node_features) or self._dropping()
        device = node_features.device
        # The reference maps the strings to ids on every call (:264-268).  Here a cached plan is used at once and, when the
        # list is too long for the cache key to cover, checked entry by entry on the host WHILE the GPU runs the forward; a
        # list edited in place fails the check: fresh plan, forward again.
        plan, check, settle = self._plan_lookup(edge_index, edge_texts, node_features.size(0), device, training=grad,
                                                defer_check=not grad)
        if grad:
            return self._forward_recorded(node_features, plan, edge_index)
        # (the check runs on the plan cache's threads, outside the GIL, beside this thread's launches and its wait for the
        # range-guard word; its result is collected before the output is handed back)
        pending = None if check is None else check_pool().submit(check)
        out = self.forward_planned(node_features, plan)
        if settle is not None:
            settle()
        if pending is not None and not pending.result():
            plan = self._plan_lookup(edge_index, edge_texts, node_features.size(0), device, training=False)[0]
            out = self.forward_planned(node_features, plan)
        return out

    # -- range guard of the two-fp16-piece kernels (include/ghf.h: ghf_set_range_flag) ---------------------------
    def _guarded(self, plan: GraphPlan) -> bool:
        pieces = plan.wlayout in _native.SPLIT_LAYOUTS or (plan.block_nodes == 1 and _native.rs_supported(self.hidden_dim)
                                                          and not _native.rs_exact(plan))
        return pieces and _native.range_guard_enabled()

    def _forward_exact(self, x: torch.Tensor, plan: GraphPlan, flags: int) -> torch.Tensor:
        """The forward again on the exact fp32 kernels: some row of h (flags & 1) or some relation's generated weights
        (flags & 2) spans more dynamic range than two fp16 pieces hold (the reference computes in plain fp32,
        hypergnn.py:202,228)."""
        self.last_range_flags = flags
        return self.forward_planned(x, exact_plan(plan, self.hidden_dim), guard=False)

    def _dropping(self) -> bool:
        """Training mode with dropout > 0 (reference :293-294): the forward then takes the recorded path — the layer's tail
        alone with a mask operand — whether or not gradients are wanted."""
        return self.training and self.dropout > 0.0

    def _draw_mask(self, shape, device) -> torch.Tensor:
        """A dropout mask scaled by 1/(1-p), drawn with torch's generator as F.dropout does in the reference."""
        return draw_mask(shape, device, self.dropout)

    def _forward_recorded(self, node_features: torch.Tensor, plan: GraphPlan, edge_index: torch.Tensor, exact: bool = False) -> torch.Tensor:
        """The forward when gradients are required (reference: plain autograd, demo.py:79-101): the same kernels inside
        ``autograd`` Functions whose backward is C-ABI calls too.  The reversed-graph plan and the relation grouping the
        backward needs are built once per plan.  When the range guard fires, the forward is recorded again on the plan for the
        exact fp32 kernels (exact=True; the first recording is dropped) — as the inference forward reruns, and as the
        reference's plain fp32 autograd needs no such thing (hypergnn.py:202,228)."""
        from ..autograd import InputProjFn, MessageLayerFn, build_train_plan
        device = node_features.device
        if plan.train is None:
            if exact:                      # (the exact plan was built from the first plan's sorted edges: take the triple from it)
                src, dst, rel = plan.edge_arrays()
                plan.train = build_train_plan(torch.stack([src, dst]), rel, plan, self.hidden_dim, device, exact=True)
            else:
                plan.train = build_train_plan(edge_index, plan.rel_ids, plan, self.hidden_dim, device)
        guard = not exact and self._guarded(plan)
        if guard:
            flag = _native.range_flag(device)
            flag.zero_()
        text_embs = self.text_encoder(plan.unique_texts, device)
        # Large graphs, no dropout: the generators on a side stream — autograd runs a Function's backward
        # on the stream of its forward, so the generators' backward (a few dozen small latency-bound kernels per layer) then
        # runs beside the message layers' gradient kernels instead of between them (C3: backward 35.2 -> 34.1 ms).  In the
        # forward the caller's stream waits for them at once: side by side with the input projection both got slower (forward
        # 13.45 -> 13.8 ms).  (With dropout the masks are drawn in the reference's order on one stream.)
        side = (plan.E >= self.SIDE_STREAM_MIN_EDGES and not self._dropping() and os.environ.get("GHF_TRAIN_WG_SIDE", "1") != "0")
        generated = []
        if side:
            main = torch.cuda.current_stream(device)
            if self._wg_stream is None or self._wg_stream[0].device != device:
                self._wg_stream = [torch.cuda.Stream(device=device) for _ in range(self.num_layers)]
            st = self._wg_stream[0]                        # one stream, one hand-over each way (a hop costs ~40 us)
            st.wait_stream(main)
            with torch.cuda.stream(st):
                for gen in self.weight_generators:
                    generated.append(gen.generate_with_grad(text_embs))
            main.wait_stream(st)
            for ws in generated:
                for t in ws:
                    t.record_stream(main)
        h = InputProjFn.apply(node_features, self.input_proj.weight, self.input_proj.bias, plan.train)
        for l, (gen, norm) in enumerate(zip(self.weight_generators, self.layer_norms)):
            if side:
                W_msg, W_self, bias = generated[l]
            else:
                W_msg, W_self, bias = gen.generate_with_grad(text_embs)
            drop = self._draw_mask(tuple(h.shape), device) if self._dropping() else None
            h = MessageLayerFn.apply(h, W_msg, W_self, bias, norm.weight, norm.bias, norm.eps, plan.train, drop)
        if guard:
            bits = int(flag.item())
            self.last_range_flags = bits
            if bits:
                # (dropout: the masks of the second recording are fresh draws — one more forward's worth of the generator)
                return self._forward_recorded(node_features, exact_plan(plan, self.hidden_dim), edge_index, exact=True)
        return h

    def generate_batched(self, text_embs: torch.Tensor, layout: int):
        """Every layer's (W_msg | Wfrag, W_self | None, bias) from ONE launch sequence (ghf_weightgen_fwd_batched: the layers'
        generators have identical shapes) — three kernels for all layers on the caller's stream: ~0.11 ms whatever the
        number of layers, where layer-by-layer generation cost that per layer (between the message launches of a small
        graph: BASELINE config 2) or needed side streams and events to hide (config 3)."""
        g0 = self.weight_generators[0]
        return _native.weightgen_fwd_batched(text_embs, [g._head_params() for g in self.weight_generators],
                                             [g._log_scale_vector() for g in self.weight_generators], g0.text_dim, g0.hidden_dim,
                                             g0.num_hidden, g0.d_in, g0.d_out, layout)

    def generate_all(self, text_embs: torch.Tensor, layout: int, side_stream: bool = True, after=None, first=None):
        """([weights of layer l], [event l or None]): every layer's weight generation.  The generated weights depend on
        the relation strings only, so on large graphs their ~0.13 ms of small latency-bound kernels per layer are
        launched on a side stream and run in the shadow of the previous layers (C3: 13.0 -> 12.7 ms per forward); the
        caller's stream waits for event l before layer l.  On small graphs the cross-stream events cost more than they
        hide (C2: 0.66 -> 0.75 ms), so there the kernels stay in the caller's stream."""
        if not side_stream:
            return [gen.generate(text_embs, layout) for gen in self.weight_generators], [None] * self.num_layers
        dev = text_embs.device
        main = torch.cuda.current_stream(dev)
        # one side stream per layer: the generators are independent chains of small latency-bound kernels, so side by side
        # all of them finish in the shadow of the input projection; in ONE side stream the later layers' kernels ran beside
        # the first message launch, which holds every CU's LDS — they trickled in as workgroups retired (0.3 ms each instead
        # of 0.05) and cost that launch 6 %
        nside = max(1, int(os.environ.get("GHF_WG_STREAMS", str(self.num_layers))))
        if self._wg_stream is None or self._wg_stream[0].device != dev or len(self._wg_stream) != nside:
            # (high-priority streams measured worse: 10.8 -> 11.6 ms per C3 forward)
            self._wg_stream = [torch.cuda.Stream(device=dev) for _ in range(nside)]
        weights, ready = [], []
        nfirst = 0 if first is None else 1
        if nfirst:                                             # layer 0's weights: already generated on the caller's stream
            weights.append(first)
            ready.append(None)
        try:
            for l, gen in enumerate(self.weight_generators):
                if l < nfirst:
                    continue
                side = self._wg_stream[l % nside]
                if l < nside + nfirst:                         # a side stream's first use in this call
                    if after is not None:
                        side.wait_event(after)                            # an event recorded once text_embs was enqueued
                    else:
                        side.wait_stream(main)                            # text_embs, and the previous call's readers
                torch.cuda.set_stream(side)
                weights.append(gen.generate(text_embs, layout))
                ev = torch.cuda.Event()
                ev.record(side)
                ready.append(ev)
        finally:
            torch.cuda.set_stream(main)
        for ws in weights:
            for t in ws:
                if t is not None:
                    t.record_stream(main)
        return weights, ready

    def forward_planned(self, node_features: torch.Tensor, plan: GraphPlan,
                        exchange=None, guard: bool = True) -> torch.Tensor:
        """Forward with an explicit plan.  `exchange(h)` (multi-GPU) runs after every layer to
        make all rows of h visible on this rank; the plan's row range says which rows it computes.
        Range guard: the kernels that cut rows / weights into two fp16 pieces flag inputs whose dynamic range those do not
        hold; the forward then ends with one 4-byte read of that flag (the only host sync of a warm forward;
        GHF_RANGE_GUARD=0 removes it) and, if it is set, runs again on the exact fp32 kernels."""
        require_inference(self, node_features, what=".forward_planned")
        if self._dropping():
            raise NotImplementedError("HyperGNN.forward_planned: dropout in training mode runs through forward() / forward_ids() "
                                      "(the recorded path); call .eval() for inference")
        device = node_features.device
        x = node_features if node_features.dtype == torch.float32 else node_features.float()
        guard = guard and exchange is None and self._guarded(plan) and not torch.cuda.is_current_stream_capturing()
        reader = None
        if guard:
            flag = _native.range_flag(device)
            flag.zero_()
            reader = _native.RangeFlagRead(flag)
        early = reader is not None and os.environ.get("GHF_GUARD_EARLY", "1") != "0"           # (0: read at the end, for A/B)
        out = self._forward_planned(x, plan, exchange, before_last=reader.arm if early else None)
        if guard:
            bits = reader.value()
            self.last_range_flags = bits
            if bits:
                return self._forward_exact(x, plan, bits)
        return out

    def _forward_planned(self, x: torch.Tensor, plan: GraphPlan, exchange=None, before_last=None) -> torch.Tensor:
        """before_last(): called right before the last layer is enqueued (nothing after that point raises a range-guard
        bit on the block kernels: _native.RangeFlagRead); the wide-row path cuts rows inside its last layer and is not
        given an early point."""
        device = x.device
        if plan.block_nodes == 1 and _native.rs_supported(self.hidden_dim):
            return self._forward_wide(x, plan, exchange)
        text_embs = self.text_encoder(plan.unique_texts, device)     # [U, text_dim]
        # the 16-bit-piece kernels gather rows already cut into pieces: the input projection emits them for the first
        # layer, every layer's tail for the next
        main = torch.cuda.current_stream(device)
        te_done = torch.cuda.Event()
        te_done.record(main)
        split = plan.wlayout in _native.SPLIT_LAYOUTS
        hs = _native.alloc_split(x.size(0), self.hidden_dim, plan.wlayout, device) if split else None
        hs_next = torch.empty_like(hs) if split else None
        # Only layer 0's weights are needed before the first message launch.  Its generator — five small latency-bound
        # kernels, ~0.15 ms alone — runs FIRST, on this stream: beside the input projection, which saturates HBM, the same
        # kernels took 0.4 + 0.3 ms and the first message launch waited for them (kernel-trace timeline, round 3: 0.76 ms
        # before the first message launch).  The later layers' generators run beside the projection on side streams as before.
        side = plan.E >= self.SIDE_STREAM_MIN_EDGES
        batched = self.num_layers <= 8 and os.environ.get("GHF_GEN_BATCHED", "1") != "0"
        if batched:                                # all layers' generators in one launch sequence, first (round 3)
            weights, ready = self.generate_batched(text_embs, plan.wlayout), [None] * self.num_layers
        w0 = self.weight_generators[0].generate(text_embs, plan.wlayout) if not batched and side and os.environ.get("GHF_GEN0_FIRST", "1") != "0" else None
        h = _native.input_proj_fwd(x, self.input_proj.weight.detach(), self.input_proj.bias.detach(), h_split=hs,
                                   split_layout=plan.wlayout if split else 0)
        h_next = torch.empty_like(h)
        # (enqueued after the input projection: streams can share a hardware queue, and packets queue in host order)
        if not batched:
            weights, ready = self.generate_all(text_embs, plan.wlayout, side_stream=side, after=te_done, first=w0)
        lo, hi = plan.row_lo, (plan.row_hi or plan.N)
        last = len(self.weight_generators) - 1
        for l, norm in enumerate(self.layer_norms):
            if ready[l] is not None:
                main.wait_event(ready[l])
            W, W_self, bias = weights[l]
            if l == last and before_last is not None:
                before_last()
            fused = split and exchange is None and l < last
            _native.message_layer_fwd(h, plan, W, W_self, bias, plan.wlayout, norm.weight.detach(),
                                      norm.bias.detach(), norm.eps, h_next, row0=lo, rows=hi - lo,
                                      h_split=hs, h_split_out=hs_next if fused else None)
            if exchange is not None:
                exchange(h_next)
                if split and l < last:
                    _native.split_rows(h_next, plan.wlayout, out=hs_next)
            h, h_next = h_next, h
            hs, hs_next = hs_next, hs
        return h

    def _forward_wide(self, x: torch.Tensor, plan: GraphPlan, exchange=None) -> torch.Tensor:
        """Wide rows (d % 128 == 0, d >= 256: BASELINE config 5): the relation-stationary layer of csrc/message_rs.hip —
        per-edge results in relation order with the weights read once per 128 edges, then destination sums + tail."""
        device = x.device
        if plan.rs is None:
            plan.rs = build_rs(plan)
        rs = plan.rs
        text_embs = self.text_encoder(plan.unique_texts, device)
        lo, hi = plan.row_lo, (plan.row_hi or plan.N)
        # rows travel between the layers already cut into fp16 pieces (written by the input projection / pass 2's tail)
        # when one process computes every row; the fp32 MFMA variant gathers h itself
        exact = _native.rs_exact(plan)
        pieces = exchange is None and lo == 0 and hi == plan.N and not exact
        hs = _native.alloc_split(plan.N, self.hidden_dim, _native.WLAYOUT_SPLIT2H, device) if pieces else None
        hs_next = torch.empty_like(hs) if pieces else None
        h = _native.input_proj_fwd(x, self.input_proj.weight.detach(), self.input_proj.bias.detach(), h_split=hs,
                                   split_layout=_native.WLAYOUT_SPLIT2H if pieces else 0)
        h_next = torch.empty_like(h)
        Y = rs.scratch(plan.E, self.hidden_dim, device)
        last = len(self.layer_norms) - 1
        all_w = self.generate_batched(text_embs, _native.WLAYOUT_NATURAL) if self.num_layers <= 8 else None
        for l, (gen, norm) in enumerate(zip(self.weight_generators, self.layer_norms)):
            W_msg, W_self, bias = all_w[l] if all_w is not None else gen.generate(text_embs, _native.WLAYOUT_NATURAL)
            if plan.E > 0:
                _native.edge_transform_fwd(h, rs, W_msg, W_self, bias, Y, h_split=hs, exact=exact)
            _native.segment_tail_fwd(Y, rs, h, norm.weight.detach(), norm.bias.detach(), norm.eps, h_next, row0=lo, rows=hi - lo,
                                     h_split_out=hs_next if pieces and l < last else None, exact=exact)
            if exchange is not None:
                exchange(h_next)
            h, h_next = h_next, h
            hs, hs_next = hs_next, hs
        return h

    # -- reference-internal seam kept for API parity (reference :160-230) ---------------------
    def _message_passing(self, h: torch.Tensor, edge_index: torch.Tensor, rel_weights: Dict[str, torch.Tensor]) -> torch.Tensor:
        """agg + self_out for per-EDGE weights ``W_msg [E,d,d]``, ``W_self [E,d,d]``, ``bias [E,d]``.

        Every edge is treated as its own relation on the generic kernel (no residual/norm);
        needs N*E < 2^32, which per-edge [E,d,d] inputs never approach."""
        require_inference(self, h, what="._message_passing")
        N, E = h.size(0), edge_index.size(1)
        rel = torch.arange(E, dtype=torch.int64, device=h.device)
        plan = build_plan(edge_index, rel, [""] * E, N, h.size(1), h.device, force_generic=True)
        out = torch.empty_like(h)
        return _native.message_layer_fwd(h, plan, rel_weights["W_msg"].contiguous(), rel_weights["W_self"].contiguous(),
                                         rel_weights["bias"].contiguous(), _native.WLAYOUT_NATURAL, None, None, 0.0,
                                         out, flags=_native.GHF_FLAG_NO_TAIL)

    # -- convenience (reference :304-322) ---------------------------------------------------
    def score_triple(self, head_emb: torch.Tensor, tail_emb: torch.Tensor) -> torch.Tensor:
        """Dot-product score of (head, tail) embeddings, ``[d]`` or ``[B, d]`` (reference :304-318)."""
        for t in (head_emb, tail_emb):
            if not t.is_cuda:
                raise RuntimeError(f"score_triple computes on an MI355X HIP device only (input is on {t.device})")
        if head_emb.shape != tail_emb.shape or head_emb.dim() not in (1, 2):
            raise ValueError(f"score_triple: shapes {tuple(head_emb.shape)} and {tuple(tail_emb.shape)}")
        single = head_emb.dim() == 1
        a = (head_emb.unsqueeze(0) if single else head_emb).float()
        b = (tail_emb.unsqueeze(0) if single else tail_emb).float()
        if torch.is_grad_enabled() and (a.requires_grad or b.requires_grad):
            from ..autograd import ScorePairsFn
            s = ScorePairsFn.apply(a, b)
        else:
            s = _native.score_pairs_fwd(a, b)
        return s[0] if single else s

    def score_edges(self, embs: torch.Tensor, src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
        """``score_triple(embs[src], embs[dst])`` (the reference's call form, demo.py:90-94) without materialising the
        two gathered ``[E, d]`` matrices — nor, when gradients are recorded, their index backward: the gradient is one
        gather pass over the pairs grouped by node (``autograd.ScoreEdgesFn``), reproducible."""
        if not embs.is_cuda:
            raise RuntimeError(f"score_edges computes on an MI355X HIP device only (input is on {embs.device})")
        if torch.is_grad_enabled() and embs.requires_grad:
            from ..autograd import ScoreEdgesFn
            return ScoreEdgesFn.apply(embs, src, dst)
        return _native.score_pairs_fwd(embs, embs, src.to(torch.int64), dst.to(torch.int64))

    def num_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)
