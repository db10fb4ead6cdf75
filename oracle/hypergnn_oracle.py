"""ORACLE — test infrastructure, NOT product code.

CPU restatement of the reference's HyperGNN forward hot path
(danieleschmidt/Graph-Hypernetwork-Forge @ /root/reference, v0.2.0), written
as plain functions over a ``{state_dict key: array}`` mapping.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package never does (it fails loudly without its HIP
library instead).

Arithmetic is stock ATen CPU (the same library the reference computes with,
``requirements.txt:2``), so this file states the *algorithm* — op order,
shapes, clamp/mean semantics — and each function cites the reference lines it
follows.  ``dtype=torch.float64`` gives a high-precision ground truth of the
same algorithm.

Parity pinning: the reference's own tests hold no numeric vectors
(SURVEY.md §4), so this oracle is pinned against outputs of the reference
itself, imported in the build container by ``tests/golden/make_golden.py`` and
committed as ``tests/golden/*.npz`` (checked by ``tests/test_oracle_golden.py``).
"""

from __future__ import annotations

import re
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

Params = Mapping[str, "np.ndarray | torch.Tensor"]

ASCII_VOCAB = 128   # reference hypergnn.py:55
LN_EPS = 1e-5       # nn.LayerNorm default, reference hypergnn.py:152-154


def _t(x, dtype: torch.dtype) -> torch.Tensor:
    if isinstance(x, torch.Tensor):                        # leaves that require grad stay attached: the backward tests
        return (x if x.requires_grad else x.detach()).to("cpu", dtype)   # differentiate this restatement with autograd
    return torch.from_numpy(np.ascontiguousarray(x)).to(dtype)


# --------------------------------------------------------------------------
# TextEncoder — reference hypergnn.py:39-81
# --------------------------------------------------------------------------

def tokenize(text: str) -> List[int]:
    """reference hypergnn.py:66-71: min(ord(c), 127) per char; '' -> [0]."""
    codes = [min(ord(c), ASCII_VOCAB - 1) for c in text]
    return codes if codes else [0]


def text_encode(params: Params, texts: Sequence[str], dtype=torch.float32) -> torch.Tensor:
    """reference hypergnn.py:73-81: mean of char embeddings -> Linear -> Tanh, per string."""
    emb = _t(params["text_encoder.char_emb.weight"], dtype)
    w = _t(params["text_encoder.proj.0.weight"], dtype)
    b = _t(params["text_encoder.proj.0.bias"], dtype)
    rows = []
    for s in texts:
        ids = torch.tensor(tokenize(s), dtype=torch.long)
        pooled = emb[ids].mean(dim=0)                     # :76
        rows.append(torch.tanh(pooled @ w.t() + b))       # :61-64, :77
    return torch.stack(rows, dim=0)                       # :81


# --------------------------------------------------------------------------
# WeightGenerator — reference weight_generator.py:96-143
# --------------------------------------------------------------------------

_HEADS = (("W_msg", 2), ("W_self", 2), ("bias", 1))      # weight_generator.py:72-76


def _head_linears(params: Params, prefix: str, head: str) -> List[Tuple[str, str]]:
    """Linear layers of one head in Sequential order.

    The Sequential indices are 0,2,4 (or 0,3,6 when Dropout modules are
    present, weight_generator.py:100-106), so they are discovered from the keys.
    """
    pat = re.compile(re.escape(f"{prefix}generators.{head}.") + r"(\d+)\.weight$")
    idx = sorted(int(m.group(1)) for k in params for m in [pat.match(k)] if m)
    return [(f"{prefix}generators.{head}.{i}.weight", f"{prefix}generators.{head}.{i}.bias") for i in idx]


def weight_generator(params: Params, prefix: str, text_emb, d_in: int, d_out: int,
                     dtype=torch.float32, drop=None) -> Dict[str, torch.Tensor]:
    """reference weight_generator.py:120-143 (eval mode: Dropout is identity).

    ``text_emb`` [T] or [B,T] -> {"W_msg": (B,)d_in x d_out, "W_self": same, "bias": (B,)d_out}.
    Training mode with dropout p > 0 (:96-107, Linear -> ReLU -> Dropout): ``drop`` [3 heads, hidden layers, B, hidden] holds the
    masks nn.Dropout would have drawn, already scaled by 1/(1-p) (nn.Dropout: y = x * mask / (1-p)).
    """
    x = _t(text_emb, dtype)
    single = x.dim() == 1                                  # :132-134
    if single:
        x = x.unsqueeze(0)
    out: Dict[str, torch.Tensor] = {}
    for head, rank in _HEADS:
        z = x
        lin = _head_linears(params, prefix, head)
        for li, (wk, bk) in enumerate(lin):                # _build_mlp :96-107
            z = z @ _t(params[wk], dtype).t() + _t(params[bk], dtype)
            if li + 1 < len(lin):
                z = torch.relu(z)
                if drop is not None:
                    z = z * _t(drop[_HEADS.index((head, rank))][li], dtype)
        scale = _t(params[f"{prefix}log_scales.{head}"], dtype).exp()   # :139
        shape = (d_in, d_out) if rank == 2 else (d_out,)
        w = z.view(x.size(0), *shape) * scale              # :140
        out[head] = w.squeeze(0) if single else w          # :141
    return out


# --------------------------------------------------------------------------
# Relation ids — reference hypergnn.py:264-268
# --------------------------------------------------------------------------

def relation_ids(edge_texts: Sequence[str]) -> Tuple[List[str], np.ndarray]:
    """Unique strings in first-appearance order and the per-edge index."""
    unique = list(dict.fromkeys(edge_texts))
    lut = {t: i for i, t in enumerate(unique)}
    return unique, np.fromiter((lut[t] for t in edge_texts), dtype=np.int64, count=len(edge_texts))


# --------------------------------------------------------------------------
# Message passing — reference hypergnn.py:160-230
# --------------------------------------------------------------------------

def message_passing_reference_shaped(h: torch.Tensor, edge_index: torch.Tensor,
                                     W_msg_e: torch.Tensor, W_self_e: torch.Tensor,
                                     bias_e: torch.Tensor) -> torch.Tensor:
    """Same op sequence as the reference, per-EDGE weights [E,d,d],[E,d,d],[E,d].

    Memory is O(E d^2), as in the reference; use only where that fits.
    """
    N, d = h.shape
    src, dst = edge_index[0], edge_index[1]                # :191
    E = src.numel()
    d_out = W_msg_e.size(-1)
    msg = torch.bmm(h[src].unsqueeze(1), W_msg_e).squeeze(1) + bias_e      # :201-204
    agg = torch.zeros(N, d_out, dtype=h.dtype)
    cnt = torch.zeros(N, 1, dtype=h.dtype)
    agg.scatter_add_(0, dst.unsqueeze(1).expand(-1, d_out), msg)            # :207-210
    cnt.scatter_add_(0, dst.unsqueeze(1), torch.ones(E, 1, dtype=h.dtype))  # :211
    cnt = cnt.clamp(min=1.0)                                                # :212
    agg = agg / cnt                                                         # :213
    W_self_agg = torch.zeros(N, d, d_out, dtype=h.dtype)                    # :217
    W_self_agg.scatter_add_(0, dst.view(-1, 1, 1).expand(-1, d, d_out), W_self_e)   # :218-219
    W_self_agg = W_self_agg / cnt.unsqueeze(-1)                             # :220
    self_out = torch.bmm(h.unsqueeze(1), W_self_agg).squeeze(1)             # :228
    return agg + self_out                                                   # :230


def message_passing_factorised(h: torch.Tensor, edge_index: torch.Tensor, rel: torch.Tensor,
                               W_msg: torch.Tensor, W_self: torch.Tensor,
                               bias: torch.Tensor) -> torch.Tensor:
    """Equivalent O(E d) memory form with per-RELATION weights [R,d,d],[R,d,d],[R,d].

    out_v = (1/max(indeg_v,1)) * sum_{e=(u->v)} (h_u W_msg[r_e] + bias[r_e] + h_v W_self[r_e])
    (SURVEY.md §8a "single fused statement"; algebraically identical to
    hypergnn.py:201-230, re-associated).  Loops over relations.
    """
    N, d = h.shape
    src, dst = edge_index[0], edge_index[1]
    d_out = W_msg.size(-1)
    acc = torch.zeros(N, d_out, dtype=h.dtype)
    cnt = torch.zeros(N, dtype=h.dtype)
    cnt.index_add_(0, dst, torch.ones(dst.numel(), dtype=h.dtype))
    order = torch.argsort(rel, stable=True)
    bounds = torch.searchsorted(rel[order].contiguous(), torch.arange(W_msg.size(0) + 1))
    for r in range(W_msg.size(0)):
        e = order[bounds[r]:bounds[r + 1]]
        if e.numel() == 0:
            continue
        contrib = h[src[e]] @ W_msg[r] + bias[r] + h[dst[e]] @ W_self[r]
        acc.index_add_(0, dst[e], contrib)
    return acc / cnt.clamp(min=1.0).unsqueeze(1)


def layer_tail(h_new: torch.Tensor, h: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
               eps: float = LN_EPS, drop=None) -> torch.Tensor:
    """reference hypergnn.py:288-296: residual, ReLU, [training: dropout, ``drop`` = the mask scaled by 1/(1-p)], LayerNorm
    (biased variance)."""
    if h_new.shape == h.shape:                             # :289-290
        h_new = h_new + h
    h_new = torch.relu(h_new)                              # :291
    if drop is not None:                                   # :293-294  F.dropout(h_new, p) = h_new * mask / (1-p)
        h_new = h_new * _t(drop, h_new.dtype)
    return torch.nn.functional.layer_norm(h_new, (h_new.size(-1),), gamma, beta, eps)   # :296


def score_triple(head_emb: torch.Tensor, tail_emb: torch.Tensor) -> torch.Tensor:
    """reference hypergnn.py:304-318: dot-product link score of [d] or [B, d] embeddings."""
    return (head_emb * tail_emb).sum(dim=-1)               # :318


# --------------------------------------------------------------------------
# Whole forward — reference hypergnn.py:236-298
# --------------------------------------------------------------------------

def num_layers_of(params: Params) -> int:
    return 1 + max(int(k.split(".")[1]) for k in params if k.startswith("layer_norms."))


def forward(params: Params, node_features, edge_index, edge_texts: Sequence[str], *,
            variant: str = "reference", dtype=torch.float32,
            return_intermediates: bool = False, drop=None):
    """HyperGNN.forward in eval mode — or, with ``drop`` = {"layers": [mask [N,d] per layer], "gen": [masks [3,nh,R,Hh] per
    layer]} (each scaled by 1/(1-p)), in training mode with the dropout masks the reference would have drawn.

    variant "reference": per-edge weight gather + the reference op sequence
    (hypergnn.py:281-286); "factorised": per-relation loop, no [E,d,d].
    """
    x = _t(node_features, dtype)
    ei = edge_index if isinstance(edge_index, torch.Tensor) else torch.from_numpy(np.asarray(edge_index))
    ei = ei.to(torch.long)
    if ei.size(1) != len(edge_texts):                      # :252-256
        raise ValueError(f"edge_index has {ei.size(1)} edges but edge_texts has {len(edge_texts)} entries")
    w_in = _t(params["input_proj.weight"], dtype)
    b_in = _t(params["input_proj.bias"], dtype)
    h = torch.relu(x @ w_in.t() + b_in)                    # :261
    d = h.size(1)
    unique, rel_np = relation_ids(edge_texts)              # :264-268
    rel = torch.from_numpy(rel_np)
    text_embs = text_encode(params, unique, dtype)         # :270
    inter = {"h0": h, "text_embs": text_embs, "rel_ids": rel}
    for l in range(num_layers_of(params)):                 # :272
        uw = weight_generator(params, f"weight_generators.{l}.", text_embs, d, d, dtype,   # :278
                              drop=None if drop is None else drop["gen"][l])
        if variant == "reference":
            h_new = message_passing_reference_shaped(      # :281-286
                h, ei, uw["W_msg"][rel], uw["W_self"][rel], uw["bias"][rel])
        elif variant == "factorised":
            h_new = message_passing_factorised(h, ei, rel, uw["W_msg"], uw["W_self"], uw["bias"])
        else:
            raise ValueError(variant)
        gamma = _t(params[f"layer_norms.{l}.weight"], dtype)
        beta = _t(params[f"layer_norms.{l}.bias"], dtype)
        h = layer_tail(h_new, h, gamma, beta, drop=None if drop is None else drop["layers"][l])   # :288-296
        inter[f"W_msg{l}"], inter[f"W_self{l}"], inter[f"bias{l}"] = uw["W_msg"], uw["W_self"], uw["bias"]
        inter[f"h{l + 1}"] = h
    return (h, inter) if return_intermediates else h


def score_triple(head_emb: torch.Tensor, tail_emb: torch.Tensor) -> torch.Tensor:
    """reference hypergnn.py:304-318."""
    return (head_emb * tail_emb).sum(dim=-1)
