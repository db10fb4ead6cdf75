"""Training through the HIP path: ``torch.autograd.Function`` wrappers whose forward AND backward are C-ABI calls.

The reference trains through plain autograd (``demo.py:79-101``, ``tests/test_hypergnn.py:183-226``,
``tests/test_weight_generator.py:86-106``).  Here autograd only records the chain; every gradient is computed by
``libghf_hip.so`` (``include/ghf.h``: "backward of the path").  With, per layer,

    out_v = (1/c_v) sum_{e=(u->v)} ( h_u Wm[r_e] + b[r_e] + h_v Ws[r_e] ),   h'_v = LayerNorm(ReLU(out_v + h_v))

and ``g' = dL/dh'``:

    (dpre, G, G_split, dgamma, dbeta) = ghf_tail_bwd(g', out, h)          G_v = dpre_v / c_v
    dWm[r] = sum_{e in r} h_u^T G_v      dWs[r] = sum_{e in r} h_v^T G_v      db[r] = sum_{e in r} G_v      (ghf_group_outer)
    dh = dpre + [sum_{e->v} G_v Ws[r]^T]_v + [sum_{e: src=u} G_{dst(e)} Wm[r]^T]_u
         (two ghf_message_layer_fwd passes with GHF_FLAG_RAW_SUM: transposed weights on the plan / on the reversed plan)

Exact fp32 contractions for the weight gradients, the message kernel for the h gradients (each of its two passes has one
zero half of the weights and says so: GHF_FLAG_ZERO_SRC / GHF_FLAG_ZERO_DST).  The training forward is the inference
launch with one more store: the kernel writes ``out`` beside ``h'`` (``agg_out``) and the split rows the next layer
gathers; kernels without that side output run GHF_FLAG_NO_TAIL plus ``ghf_tail_fwd``.

The callers either side of the layer have their own Functions here: ``WeightGeneratorFn`` (three MLP heads and the
learnable log-scales, reference weight_generator.py:120-143), ``InputProjFn`` (hypergnn.py:261), ``TextEncoderFn``
(hypergnn.py:57-81), ``ScorePairsFn`` (hypergnn.py:304-318) and its fused form over index arrays ``ScoreEdgesFn``.  All of
their contractions are ``A^T B`` over rows, i.e. ``ghf_group_outer`` again (``_native.matmul_tn``).
"""

from __future__ import annotations

import os
import weakref
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _native
from .plan import GraphPlan, build_plan, build_rs


@dataclass
class TrainPlan:
    """What a training step needs beyond the forward plan: the reversed graph's plan and the edges grouped by relation."""
    fwd: GraphPlan
    rev: GraphPlan
    src_by_rel: torch.Tensor      # [E] int64: source of the edges in relation order (ascending destination inside one)
    dst_by_rel: torch.Tensor      # [E] int64
    goff: torch.Tensor            # [R+1] int64
    slice_tab: Optional[torch.Tensor] = None   # [S, 3] int64 (relation, first edge, end edge): ghf_edge_outer's work list
    slice_off: Optional[torch.Tensor] = None   # [R+1] int64
    slice_order: Optional[torch.Tensor] = None # [S] int32: the slices in launch order (by position inside their relation)
    ident: Optional[tuple] = None               # (arange(N), slice table, slice offsets): InputProjFn's weight gradient as ghf_edge_outer
    zero_bias: Optional[torch.Tensor] = None    # [R, d] zeros: the bias operand of the two gradient passes
    carry: Optional["_SplitCarry"] = None       # split rows handed from one layer's launch to the next

SLICE_EDGES = 4096                # edges per ghf_edge_outer workgroup (a multiple of its 32-edge tile)


def build_train_plan(edge_index: torch.Tensor, rel_ids: torch.Tensor, fwd: GraphPlan, d: int, device, exact: bool = False) -> TrainPlan:
    """exact: `fwd` is a plan for the exact fp32 kernels (the range guard's fallback): the reversed graph is planned for them
    too, and wide rows run the relation-stationary layer on fp32 MFMAs in both directions (GraphPlan.force_exact)."""
    if fwd.row_lo != 0 or (fwd.row_hi or fwd.N) != fwd.N or fwd.E != edge_index.size(1):
        raise NotImplementedError("training runs on single-GPU plans (every destination row, every edge)")
    ei = edge_index.to(device=device, dtype=torch.int64)
    rel = rel_ids.to(device=device, dtype=torch.int64).contiguous()
    rev = build_plan(ei.flip(0).contiguous(), rel, fwd.unique_texts, fwd.N, d, device, exact=exact)
    fwd.force_exact = rev.force_exact = exact
    by_dst = torch.sort(ei[1], stable=True).indices          # destinations ascending inside a relation: G / h_dst rows stay
    perm, goff = _native.group_edges(rel.index_select(0, by_dst), fwd.R)   # hot in L2 while a slice is contracted
    perm = by_dst.index_select(0, perm)
    tp = TrainPlan(fwd=fwd, rev=rev, src_by_rel=ei[0].index_select(0, perm).contiguous(),
                   dst_by_rel=ei[1].index_select(0, perm).contiguous(), goff=goff, carry=_SplitCarry())
    if _native.load().ghf_edge_outer_supported(d):
        off = goff.cpu().tolist()                             # (one host sync per plan)
        tab, soff = [], [0]
        for r in range(fwd.R):
            tab += [(r, a, min(a + SLICE_EDGES, off[r + 1])) for a in range(off[r], off[r + 1], SLICE_EDGES)]
            soff.append(len(tab))
        if tab:
            tp.slice_tab = torch.tensor(tab, dtype=torch.int64).to(device)
            tp.slice_off = torch.tensor(soff, dtype=torch.int64).to(device)
            if _EO_ORDER:
                # Destinations ascend inside a relation, so a slice's place in its relation (as a fraction of the relation's
                # slices) says which band of destination rows it reads.  Launched band by band — every relation's slice of the
                # band together — the workgroups in flight share that band's rows of h and G (Infinity Cache) instead of each
                # relation sweeping all N rows alone.  Same bits: ghf.h, ghf_edge_outer.
                place = [((k + 0.5) / (soff[r + 1] - soff[r]), r) for r in range(fwd.R) for k in range(soff[r + 1] - soff[r])]
                order = sorted(range(len(tab)), key=place.__getitem__)
                tp.slice_order = torch.tensor(order, dtype=torch.int32).to(device)
    return tp


def _layer_weights(plan: GraphPlan, Wm: Optional[torch.Tensor], Ws: Optional[torch.Tensor], transpose: bool):
    """(W, W_self) arguments of ghf_message_layer_fwd for natural [R,d,d] matrices (None = zeros) in the plan's layout."""
    R, d = plan.R, (Wm if Wm is not None else Ws).size(1)
    if plan.wlayout == _native.WLAYOUT_NATURAL:
        z = None

        def nat(w):
            nonlocal z
            if w is None:
                if z is None:
                    z = torch.zeros(R, d, d, dtype=torch.float32, device=(Wm if Wm is not None else Ws).device)
                return z
            return _native.transpose_batched(w) if transpose else w.contiguous()
        return nat(Wm), nat(Ws)
    if plan.wlayout not in (_native.WLAYOUT_SPLIT2H, _native.WLAYOUT_FRAG16):
        raise NotImplementedError(f"training needs the generic, the FRAG16 or the SPLIT2H message kernel (plan layout "
                                  f"{plan.wlayout}; unset GHF_KERNEL)")
    return _native.weights_pack(Wm, Ws, transpose, R, d, plan.wlayout), None


def _message(x: torch.Tensor, plan: GraphPlan, W, W_self, bias: torch.Tensor, flags: int, x_split=None, residual=None) -> torch.Tensor:
    """A message pass without tail (NO_TAIL / RAW_SUM) on the plan's kernel: the destination-block or generic kernel, or —
    CSR plans of wide rows — the relation-stationary layer."""
    out = torch.empty_like(x)
    if plan.block_nodes == 1 and _native.rs_supported(x.size(1)) and plan.E > 0:
        if plan.rs is None:
            plan.rs = build_rs(plan)
        Y = plan.rs.scratch(plan.E, x.size(1), x.device)      # (plan.force_exact — the guard's fallback: pass 1 on fp32 MFMAs, forward and backward)
        _native.edge_transform_fwd(x, plan.rs, W, W_self, bias, Y, exact=plan.force_exact)
        _native.segment_tail_fwd(Y, plan.rs, None, None, None, 0.0, out, flags=flags, exact=plan.force_exact)
        return out
    if residual is not None:                 # out = (the pass) + residual, added in the kernel's tail (GHF_FLAG_ADD_H): x_split names
        _native.message_layer_fwd(residual, plan, W, W_self, bias, plan.wlayout, None, None, 0.0, out,      # the gathered rows
                                  flags=flags | _native.GHF_FLAG_ADD_H, h_split=x_split)
        return out
    _native.message_layer_fwd(x, plan, W, W_self, bias, plan.wlayout, None, None, 0.0, out, flags=flags, h_split=x_split)
    return out


def _raw_message(x: torch.Tensor, plan: GraphPlan, W, W_self, zero_bias: torch.Tensor, zero_half: int, x_split=None, residual=None) -> torch.Tensor:
    # GHF_FLAG_ZERO_*: "this half must not be read" (include/ghf.h) — the block kernels honour it; the others are handed
    # packs whose other half really is zero (MessageLayerFn.backward) and no flag
    if not _native.side_output_supported(plan, x.size(1)):
        zero_half = 0
    return _message(x, plan, W, W_self, zero_bias, _native.GHF_FLAG_RAW_SUM | zero_half, x_split, residual)


class _SplitCarry:
    """The split form of a layer's output rows, written by that layer's launch for the next layer's gathers (the
    inference path's h_split chain).  Valid for the very tensor object the launch returned, unmodified since — anything
    else (another tensor, an in-place edit) finds nothing and the rows are cut again."""

    def __init__(self) -> None:
        self.ref = None
        self.version = -1
        self.split = None

    def take(self, h: torch.Tensor):
        hit = self.ref is not None and self.ref() is h and h._version == self.version
        split, self.ref, self.split = (self.split if hit else None), None, None
        return split

    def put(self, h: torch.Tensor, split: torch.Tensor) -> None:
        self.ref, self.version, self.split = weakref.ref(h), h._version, split


# GHF_EO_SIDE=1: the weight gradients on a second stream beside the two gradient passes.  Round 2: 49.1 -> 48.7 ms per C3
# training step.  Round 3 (same box, tools/ab_train.sh): 45.0-45.9 ms with, 45.1-45.5 without — the three kernels contend for
# the same gather path (beside the contraction the self-term pass took 4.8 ms, alone 2.4): off, one stream fewer.
_EO_SIDE = os.environ.get("GHF_EO_SIDE", "0") != "0"
_WG_FUSED_BWD = int(os.environ.get("GHF_WG_FUSED_BWD", "1"))       # WeightGeneratorFn.backward through ghf_weightgen_bwd: 0 never, 1 small generators, 2 always
WG_FUSED_MAX = 1 << 18            # ... "small": relations x d_in x d_out (config 2: 2^17, config 3: 2^20)
_IP_EDGE_OUTER = os.environ.get("GHF_IP_EDGE_OUTER", "1") != "0"   # InputProjFn.backward: dW / db through ghf_edge_outer
_EO_ORDER = os.environ.get("GHF_EO_ORDER", "1") != "0"     # ghf_edge_outer's slices launched band by band (build_train_plan)
_ONE_PACK = os.environ.get("GHF_BWD_ONE_PACK", "1") != "0"   # the two gradient passes share one packed weight tensor (three pack
                                                             # launches fewer per step; within the box noise of tools/ab_train.sh)
_SIDE_STREAMS: dict = {}


def _side_stream(device) -> torch.cuda.Stream:
    key = torch.device(device).index or 0
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class MessageLayerFn(torch.autograd.Function):
    """One HyperGNN layer (reference hypergnn.py:281-296) with per-relation weights in natural layout."""

    @staticmethod
    def forward(ctx, h, W_msg, W_self, bias, gamma, beta, eps: float, tp: TrainPlan, drop: Optional[torch.Tensor] = None):
        """drop: the layer's dropout mask scaled by 1/(1-p) ([N, d]; reference hypergnn.py:293-294: between ReLU and LayerNorm),
        None without dropout."""
        plan = tp.fwd
        h = h.contiguous()
        W, W2 = _layer_weights(plan, W_msg.detach(), W_self.detach(), transpose=False)
        out = torch.empty_like(h)
        h_scales = None
        if drop is None and _native.side_output_supported(plan, h.size(1)):   # one launch: h', the aggregate, the next layer's split rows
            agg = torch.empty_like(h)
            hs = tp.carry.take(h)
            if hs is None:
                hs = _native.split_rows(h, plan.wlayout)
            hs_out = torch.empty_like(hs)
            _native.message_layer_fwd(h, plan, W, W2, bias.detach().contiguous(), plan.wlayout, gamma.detach(), beta.detach(), eps,
                                      out, h_split=hs, h_split_out=hs_out, agg_out=agg)
            tp.carry.put(out, hs_out)
            if plan.wlayout == _native.WLAYOUT_SPLIT2H:
                # 4 bytes per row: the weight gradients read their scale for h off these (ghf_edge_outer_scaled)
                h_scales = _native.split_row_scales(hs, h.size(0), h.size(1)).clone()
        else:
            agg = _message(h, plan, W, W2, bias.detach().contiguous(), _native.GHF_FLAG_NO_TAIL)
            _native.tail_fwd(agg, h, gamma.detach(), beta.detach(), eps, out, drop=drop)
        ctx.save_for_backward(h, agg, W_msg, W_self, gamma)
        ctx.tp, ctx.eps, ctx.drop, ctx.h_scales = tp, eps, drop, h_scales
        return out

    @staticmethod
    def backward(ctx, grad_out):
        h, agg, W_msg, W_self, gamma = ctx.saved_tensors
        tp: TrainPlan = ctx.tp
        plan = tp.fwd
        g = grad_out.contiguous().float()
        # both gradient passes gather the same rows of G: two-piece plans get them cut by the same launch
        split_G = (ctx.needs_input_grad[0] and plan.wlayout in _native.SPLIT_LAYOUTS and tp.rev.wlayout == plan.wlayout
                   and plan.block_nodes > 1)
        dpre, G, Gs, dgamma, dbeta = _native.tail_bwd(g, agg, h, gamma.detach(), ctx.eps, plan.indeg, drop=ctx.drop,
                                                      split_layout=plan.wlayout if split_G else None)
        side = None
        scales = {}
        if Gs is not None and ctx.h_scales is not None and plan.wlayout == _native.WLAYOUT_SPLIT2H:
            scales = dict(h_scales=ctx.h_scales, G_scales=_native.split_row_scales(Gs, G.size(0), G.size(1)))
        if tp.slice_tab is not None:
            if _EO_SIDE and ctx.needs_input_grad[0]:
                # the weight gradients (bound by their row gathers) beside the two gradient passes (bound inside the CU): two
                # streams, joined before the results leave
                main = torch.cuda.current_stream(h.device)
                side = _side_stream(h.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    dW, db = _native.edge_outer(h, G, tp.src_by_rel, tp.dst_by_rel, tp.slice_tab, tp.slice_off, plan.R, exact=plan.force_exact, order=tp.slice_order, **scales)
            else:
                dW, db = _native.edge_outer(h, G, tp.src_by_rel, tp.dst_by_rel, tp.slice_tab, tp.slice_off, plan.R, exact=plan.force_exact, order=tp.slice_order, **scales)
            d = h.size(1)
            dWm, dWs = dW[:, :d], dW[:, d:]
        else:
            dWm = _native.group_outer(h, tp.src_by_rel, G, tp.dst_by_rel, tp.goff)
            dWs = _native.group_outer(h, tp.dst_by_rel, G, tp.dst_by_rel, tp.goff)
            db = _native.group_outer(None, None, G, tp.dst_by_rel, tp.goff).reshape(plan.R, -1)
        dh = None
        if ctx.needs_input_grad[0]:
            if tp.zero_bias is None or tp.zero_bias.shape != (plan.R, h.size(1)):      # (one fill per plan, not one per layer and step)
                tp.zero_bias = torch.zeros(plan.R, h.size(1), dtype=torch.float32, device=h.device)
            zero_b = tp.zero_bias
            if _ONE_PACK and plan.wlayout == tp.rev.wlayout and plan.wlayout in _native.SPLIT_LAYOUTS:
                # one packed tensor serves both passes: each declares the half it does not read zero (ZERO_SRC / ZERO_DST)
                Wf, Wf2 = _layer_weights(plan, W_msg.detach(), W_self.detach(), transpose=True)
                Wr, Wr2 = Wf, Wf2
            else:
                Wf, Wf2 = _layer_weights(plan, None, W_self.detach(), transpose=True)       # self term: rows keyed by destination
                Wr, Wr2 = _layer_weights(tp.rev, W_msg.detach(), None, transpose=True)      # message term: scattered to the sources
            if Gs is not None and _native.side_output_supported(plan, h.size(1)) and _native.side_output_supported(tp.rev, h.size(1)):
                # the three terms are added in the two passes' tails: dpre + self term, then + message term
                t1 = _raw_message(G, plan, Wf, Wf2, zero_b, _native.GHF_FLAG_ZERO_SRC, Gs, residual=dpre)
                dh = _raw_message(G, tp.rev, Wr, Wr2, zero_b, _native.GHF_FLAG_ZERO_DST, Gs, residual=t1)
            else:
                dh = _native.add3(dpre, _raw_message(G, plan, Wf, Wf2, zero_b, _native.GHF_FLAG_ZERO_SRC, Gs),
                                  _raw_message(G, tp.rev, Wr, Wr2, zero_b, _native.GHF_FLAG_ZERO_DST, Gs), out=dpre)
        if side is not None:
            main.wait_stream(side)
            dW.record_stream(main)
            db.record_stream(main)
        return dh, dWm, dWs, db, dgamma, dbeta, None, None, None


class WeightGeneratorFn(torch.autograd.Function):
    """(W_msg [R,d_in,d_out], W_self [R,d_in,d_out], bias [R,d_out]) = exp(log_scale_k) * MLP_k(text_emb), k = three heads.

    Arguments after the dims: text_emb [R,T], the three log-scales ([1] each), then the flat parameter list
    [head][layer][weight, bias] exactly as ghf_weightgen_fwd takes it."""

    @staticmethod
    def forward(ctx, dims, x, ls0, ls1, ls2, *params):
        """dims = (T, Hh, nh, d_in, d_out[, hidden_drop, log_keep]): hidden_drop = the hidden layers' dropout masks scaled by
        1/(1-p) ([3, nh, R, Hh]; reference weight_generator.py:96-107) and log_keep = a 1-element device tensor log(1/(1-p))."""
        T, Hh, nh, d_in, d_out = dims[:5]
        hidden_drop, log_keep = (dims[5], dims[6]) if len(dims) > 5 else (None, None)
        x = x.contiguous().float()
        flat = [p.detach() for p in params]
        ls = torch.cat([ls0.detach().reshape(1), ls1.detach().reshape(1), ls2.detach().reshape(1)])
        # (the hidden activations the backward needs leave with the same launch)
        *outs, acts = _native.weightgen_fwd(x, flat, ls, T, Hh, nh, d_in, d_out, _native.WLAYOUT_NATURAL, hidden_drop=hidden_drop,
                                            want_acts=True)
        outs = tuple(outs)
        ctx.dims, ctx.acts, ctx.n_params, ctx.log_keep = dims[:5], acts, len(params), log_keep
        ctx.save_for_backward(x, ls, *outs, *params)
        return outs

    @staticmethod
    def backward(ctx, *grads):
        T, Hh, nh, d_in, d_out = ctx.dims
        x, ls = ctx.saved_tensors[0], ctx.saved_tensors[1]
        outs, params = ctx.saved_tensors[2:5], ctx.saved_tensors[5:]
        R, nl = x.size(0), nh + 1
        if (_WG_FUSED_BWD and (_WG_FUSED_BWD > 1 or R * d_in * d_out <= WG_FUSED_MAX) and all(g is not None for g in grads)
                and _native.weightgen_bwd_supported(T, Hh, nh)):
            # all heads, all layers: three launches (csrc/weightgen_bwd.hip) instead of ~68.  For small generators only: the
            # three kernels are chains of short latency-bound phases (0.1 ms per call at BASELINE configs 1 and 2, 0.25 ms at
            # config 3) — a win where the step is launch latencies (config 1: 4.7 -> 2.4 ms per step), a loss where the ~68
            # small launches hide on the side stream beside the gradient passes (config 3: 42.1 -> 42.9 ms).
            dparams, dls3, dx = _native.weightgen_bwd(x, [p.detach() for p in params], ctx.acts, [o.view(R, -1) for o in outs],
                                                      [g.contiguous().float().view(R, -1) for g in grads], ls, T, Hh, nh, d_in, d_out,
                                                      log_keep=ctx.log_keep, want_dx=ctx.needs_input_grad[1])
            return (None, dx, dls3[0:1], dls3[1:2], dls3[2:3], *dparams)
        dls: List[Optional[torch.Tensor]] = []
        dparams: List[Optional[torch.Tensor]] = [None] * len(params)
        dxs: List[torch.Tensor] = []
        for k in range(3):
            if grads[k] is None:
                dls.append(None)
                continue
            g = grads[k].contiguous().float().view(R, -1)
            dls.append(_native.dot(g, outs[k]))                        # out = y exp(ls): d out / d ls = out
            dy = _native.scale_exp(g, ls[k:k + 1])
            for l in range(nl - 1, -1, -1):
                W = params[(k * nl + l) * 2].detach()
                a_prev = ctx.acts[k, l - 1] if l > 0 else x
                if l < nl - 1:
                    # acts are post-dropout: a > 0 <=> kept and active; a kept unit's gradient carries the mask's 1/(1-p)
                    dy = _native.relu_mask(dy, ctx.acts[k, l])
                    if ctx.log_keep is not None:
                        dy = _native.scale_exp(dy, ctx.log_keep)
                dparams[(k * nl + l) * 2] = _native.matmul_tn(dy, a_prev)
                dparams[(k * nl + l) * 2 + 1] = _native.colsum(dy)
                if l > 0 or ctx.needs_input_grad[1]:
                    dy = _native.matmul_nn(dy, W)
            dxs.append(dy)
        dx = None
        if ctx.needs_input_grad[1] and dxs:
            dx = dxs[0] if len(dxs) == 1 else _native.add3(dxs[0], dxs[1], dxs[2] if len(dxs) > 2 else None)
        return (None, dx, *dls, *dparams)


class InputProjFn(torch.autograd.Function):
    """h0 = relu(x W^T + b) (reference hypergnn.py:261)."""

    @staticmethod
    def forward(ctx, x, W, b, tp=None):
        """tp (a TrainPlan on a two-piece layout): the rows also leave cut into pieces for the first layer's gathers, by
        the same launch, through the plan's carry."""
        x = x.contiguous().float()
        split = tp is not None and tp.fwd.wlayout in _native.SPLIT_LAYOUTS and tp.fwd.block_nodes > 1
        hs = _native.alloc_split(x.size(0), W.size(0), tp.fwd.wlayout, x.device) if split else None
        h0 = _native.input_proj_fwd(x, W.detach(), b.detach(), h_split=hs, split_layout=tp.fwd.wlayout if split else 0)
        if split:
            tp.carry.put(h0, hs)
        ctx.save_for_backward(x, W, h0)
        ctx.tp = tp
        return h0

    @staticmethod
    def backward(ctx, g):
        x, W, h0 = ctx.saved_tensors
        dz = _native.relu_mask(g.contiguous().float(), h0)
        tp, N, d = ctx.tp, x.size(0), x.size(1)
        if (_IP_EDGE_OUTER and tp is not None and W.size(0) == d and N >= 65536 and _native.load().ghf_edge_outer_supported(d)):
            # dW^T = x^T dz and db = the column sums of dz are ghf_edge_outer's sums over the "edges" v -> v of one relation
            # (its [h_src | h_dst] is [x | x]: the second read of a row is an L2 hit): the tall contraction on the matrix pipe
            # with its slices over all CUs instead of ghf_group_outer + two ghf_colsum (1.0 -> 0.3 ms at C3).  Same two-piece
            # arithmetic and one-scale-per-tensor contract as the layers' weight gradients (ghf.h); the exact chain when the
            # step fell back to the exact kernels.
            if tp.ident is None or tp.ident[0].numel() != N:
                step = min(SLICE_EDGES, max(256, (N // 1024 + 31) // 32 * 32))       # ~1,000 slices: all CUs also at 10^5 rows
                tab = [(0, a, min(a + step, N)) for a in range(0, N, step)]
                tp.ident = (torch.arange(N, dtype=torch.int64, device=x.device), torch.tensor(tab, dtype=torch.int64).to(x.device),
                            torch.tensor([0, len(tab)], dtype=torch.int64).to(x.device))
            ids, tab, soff = tp.ident
            dWt, db1 = _native.edge_outer(x, dz, ids, ids, tab, soff, 1, exact=tp.fwd.force_exact)
            dW, db = dWt[0, :d].t().contiguous(), db1[0]
        else:
            dW = _native.matmul_tn(dz, x)
            db = _native.colsum(dz)
        dx = None
        if ctx.needs_input_grad[0]:
            # dz W in slabs of rows: one launch holds at most 65,535 x 16 rows of the result (a grid dimension)
            Wc, slab = W.detach().contiguous(), 16 * 65535
            if x.size(0) <= slab:
                dx = _native.matmul_nn(dz, Wc)
            else:
                dx = torch.empty(x.size(0), Wc.size(1), dtype=torch.float32, device=x.device)
                for a in range(0, x.size(0), slab):
                    dx[a:a + slab] = _native.matmul_nn(dz[a:a + slab], Wc)
        return dx, dW, db, None


class TextEncoderFn(torch.autograd.Function):
    """te = tanh(mean_l E[ids] W^T + b) for all strings at once (reference hypergnn.py:57-81)."""

    @staticmethod
    def forward(ctx, E, W, b, ids, lens):
        te = _native.text_encode_fwd(ids, lens, E.detach(), W.detach(), b.detach())
        ctx.save_for_backward(E, W, te, ids, lens)
        return te

    @staticmethod
    def backward(ctx, g):
        E, W, te, ids, lens = ctx.saved_tensors
        dE, dW, db = _native.text_encode_bwd(ids, lens, E.detach(), W.detach(), te, g.contiguous().float())
        return dE, dW, db, None, None


class ScoreEdgesFn(torch.autograd.Function):
    """s_i = embs[src_i] . embs[dst_i] without the two gathered [P, d] matrices (HyperGNN.score_edges).  Backward: the 2P
    (node, partner, pair) entries grouped by node (ghf_group_edges, stable) and summed per node in that order —
    d embs[v] = sum_{i: src_i = v} g_i embs[dst_i] + sum_{i: dst_i = v} g_i embs[src_i] — one gather pass, reproducible."""

    @staticmethod
    def forward(ctx, embs, src, dst):
        embs = embs.contiguous().float()
        src, dst = src.to(torch.int64).contiguous(), dst.to(torch.int64).contiguous()
        ctx.save_for_backward(embs, src, dst)
        return _native.score_pairs_fwd(embs, embs, src, dst)

    @staticmethod
    def backward(ctx, g):
        embs, src, dst = ctx.saved_tensors
        P, N = src.numel(), embs.size(0)
        keys = torch.cat([src, dst]).clamp_(0, N - 1)
        perm, off = _native.group_edges(keys, N)
        partner = torch.cat([dst, src]).index_select(0, perm)
        pair = torch.remainder(perm, P)
        return _native.segment_axpy(g.contiguous().float(), pair, embs, partner, off), None, None


class ScorePairsFn(torch.autograd.Function):
    """s_i = a_i . b_i (reference score_triple, hypergnn.py:304-318)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous().float(), b.contiguous().float()
        ctx.save_for_backward(a, b)
        return _native.score_pairs_fwd(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous().float()
        return (_native.rowscale(b, g) if ctx.needs_input_grad[0] else None,
                _native.rowscale(a, g) if ctx.needs_input_grad[1] else None)
