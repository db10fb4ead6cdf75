"""Training through the HIP path: ``torch.autograd.Function`` wrappers whose forward AND backward are C-ABI calls.

The reference trains through plain autograd (``demo.py:79-101``, ``tests/test_hypergnn.py:183-226``,
``tests/test_weight_generator.py:86-106``).  Here autograd only records the chain; every gradient is computed by
``libghf_hip.so`` (``include/ghf.h``: "backward of the path").  With, per layer,

    out_v = (1/c_v) sum_{e=(u->v)} ( h_u Wm[r_e] + b[r_e] + h_v Ws[r_e] ),   h'_v = LayerNorm(ReLU(out_v + h_v))

and ``g' = dL/dh'``:

    (dpre, G, T) = ghf_tail_bwd(g', out, h)          G_v = dpre_v / c_v
    dgamma = colsum(T), dbeta = colsum(g')
    dWm[r] = sum_{e in r} h_u^T G_v      dWs[r] = sum_{e in r} h_v^T G_v      db[r] = sum_{e in r} G_v      (ghf_group_outer)
    dh = dpre + [sum_{e->v} G_v Ws[r]^T]_v + [sum_{e: src=u} G_{dst(e)} Wm[r]^T]_u
         (two ghf_message_layer_fwd passes with GHF_FLAG_RAW_SUM: transposed weights on the plan / on the reversed plan)

First version: exact fp32 contractions for the weight gradients, the message kernel for the h gradients; the
training forward runs the message kernel with GHF_FLAG_NO_TAIL plus ``ghf_tail_fwd`` so that ``out`` can be saved.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import _native
from .plan import GraphPlan, build_plan


@dataclass
class TrainPlan:
    """What a training step needs beyond the forward plan: the reversed graph's plan and the edges grouped by relation."""
    fwd: GraphPlan
    rev: GraphPlan
    src_by_rel: torch.Tensor      # [E] int64: source of the edges in relation order
    dst_by_rel: torch.Tensor      # [E] int64
    goff: torch.Tensor            # [R+1] int64


def build_train_plan(edge_index: torch.Tensor, rel_ids: torch.Tensor, fwd: GraphPlan, d: int, device) -> TrainPlan:
    ei = edge_index.to(device=device, dtype=torch.int64)
    rel = rel_ids.to(device=device, dtype=torch.int64).contiguous()
    rev = build_plan(ei.flip(0).contiguous(), rel, fwd.unique_texts, fwd.N, d, device)
    perm, goff = _native.group_edges(rel, fwd.R)
    return TrainPlan(fwd=fwd, rev=rev, src_by_rel=ei[0].index_select(0, perm).contiguous(),
                     dst_by_rel=ei[1].index_select(0, perm).contiguous(), goff=goff)


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
    if plan.wlayout != _native.WLAYOUT_SPLIT2H:
        raise NotImplementedError(f"training needs the generic or the SPLIT2H message kernel (plan layout {plan.wlayout})")
    return _native.weights_pack(Wm, Ws, transpose, R, d, plan.wlayout), None


def _raw_message(x: torch.Tensor, plan: GraphPlan, W, W_self, zero_bias: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(x)
    _native.message_layer_fwd(x, plan, W, W_self, zero_bias, plan.wlayout, None, None, 0.0, out,
                              flags=_native.GHF_FLAG_RAW_SUM)
    return out


class MessageLayerFn(torch.autograd.Function):
    """One HyperGNN layer (reference hypergnn.py:281-296) with per-relation weights in natural layout."""

    @staticmethod
    def forward(ctx, h, W_msg, W_self, bias, gamma, beta, eps: float, tp: TrainPlan):
        plan = tp.fwd
        h = h.contiguous()
        W, W2 = _layer_weights(plan, W_msg.detach(), W_self.detach(), transpose=False)
        agg = torch.empty_like(h)
        _native.message_layer_fwd(h, plan, W, W2, bias.detach().contiguous(), plan.wlayout, None, None, 0.0, agg,
                                  flags=_native.GHF_FLAG_NO_TAIL)
        out = torch.empty_like(h)
        _native.tail_fwd(agg, h, gamma.detach(), beta.detach(), eps, out)
        ctx.save_for_backward(h, agg, W_msg, W_self, gamma)
        ctx.tp, ctx.eps = tp, eps
        return out

    @staticmethod
    def backward(ctx, grad_out):
        h, agg, W_msg, W_self, gamma = ctx.saved_tensors
        tp: TrainPlan = ctx.tp
        plan = tp.fwd
        g = grad_out.contiguous().float()
        dpre, G, T = _native.tail_bwd(g, agg, h, gamma.detach(), ctx.eps, plan.indeg)
        dgamma = _native.colsum(T)
        dbeta = _native.colsum(g)
        dWm = _native.group_outer(h, tp.src_by_rel, G, tp.dst_by_rel, tp.goff)
        dWs = _native.group_outer(h, tp.dst_by_rel, G, tp.dst_by_rel, tp.goff)
        db = _native.group_outer(None, None, G, tp.dst_by_rel, tp.goff).reshape(plan.R, -1)
        dh = None
        if ctx.needs_input_grad[0]:
            zero_b = torch.zeros(plan.R, h.size(1), dtype=torch.float32, device=h.device)
            Wf, Wf2 = _layer_weights(plan, None, W_self.detach(), transpose=True)       # self term: rows keyed by destination
            Wr, Wr2 = _layer_weights(tp.rev, W_msg.detach(), None, transpose=True)      # message term: scattered to the sources
            dh = dpre + _raw_message(G, plan, Wf, Wf2, zero_b) + _raw_message(G, tp.rev, Wr, Wr2, zero_b)
        return dh, dWm, dWs, db, dgamma, dbeta, None, None
