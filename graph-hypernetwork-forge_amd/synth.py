"""Build-owned synthetic inputs: counter-based RNG, synthetic KGs, parameter init.

Everything here is a pure function of (seed, stream name, flat index), so this
container, the GPU box and the golden-fixture generator all produce the same
graphs and parameters without shipping them (SURVEY.md §8c G5, §8d "Synthetic
inputs").  Integers are bit-identical everywhere; normals are computed in
float64 and rounded to float32.

The generator is splitmix64 applied to a counter; nothing here comes from the
reference (which has no data generator beyond ``torch.randn`` in
``graph_hypernetwork_forge/data/knowledge_graph.py:75-79``).
"""

from __future__ import annotations

import math
import zlib
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

_U64 = np.uint64
_GOLD = _U64(0x9E3779B97F4A7C15)
_M1 = _U64(0xBF58476D1CE4E5B9)
_M2 = _U64(0x94D049BB133111EB)


def _mix(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z + _GOLD).astype(_U64)
        z = ((z ^ (z >> _U64(30))) * _M1).astype(_U64)
        z = ((z ^ (z >> _U64(27))) * _M2).astype(_U64)
        return (z ^ (z >> _U64(31))).astype(_U64)


def stream_id(name: str) -> int:
    return zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF


def raw_u64(seed: int, name: str, n: int, offset: int = 0) -> np.ndarray:
    """n raw 64-bit words for stream `name`, counters offset..offset+n-1."""
    base = _mix(np.array([(seed << 32) ^ stream_id(name)], dtype=_U64))[0]
    idx = np.arange(offset, offset + n, dtype=_U64)
    with np.errstate(over="ignore"):
        return _mix((idx * _GOLD + base).astype(_U64))


def uniform01(seed: int, name: str, n: int) -> np.ndarray:
    """float64 uniforms in [0, 1) with 53 random bits."""
    return (raw_u64(seed, name, n) >> _U64(11)).astype(np.float64) * (1.0 / (1 << 53))


def randint(seed: int, name: str, n: int, high: int) -> np.ndarray:
    """int64 in [0, high) by multiply-shift on the top 32 bits."""
    hi = (raw_u64(seed, name, n) >> _U64(32)).astype(_U64)
    return ((hi * _U64(high)) >> _U64(32)).astype(np.int64)


def normal(seed: int, name: str, shape: Tuple[int, ...], std: float = 1.0) -> np.ndarray:
    """float32 N(0, std^2) by Box-Muller on two uniforms per value."""
    n = int(np.prod(shape)) if len(shape) else 1
    w = raw_u64(seed, name, 2 * n)
    u1 = ((w[:n] >> _U64(11)).astype(np.float64) + 1.0) * (1.0 / (1 << 53))  # (0, 1]
    u2 = (w[n:] >> _U64(11)).astype(np.float64) * (1.0 / (1 << 53))
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)
    return (z * std).astype(np.float32).reshape(shape)


def uniform(seed: int, name: str, shape: Tuple[int, ...], bound: float) -> np.ndarray:
    """float32 U(-bound, bound)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, name, n)
    return ((2.0 * u - 1.0) * bound).astype(np.float32).reshape(shape)


# ---------------------------------------------------------------------------
# Synthetic knowledge graphs
# ---------------------------------------------------------------------------

@dataclass
class SynthKG:
    """A synthetic KG in the form ``HyperGNN.forward`` takes it."""

    node_features: np.ndarray      # [N, F] float32
    edge_index: np.ndarray         # [2, E] int64 (row 0 = src, row 1 = dst)
    rel_ids: np.ndarray            # [E] int64 index into relation_texts
    relation_texts: List[str]      # R strings

    @property
    def num_nodes(self) -> int:
        return self.node_features.shape[0]

    @property
    def num_edges(self) -> int:
        return self.edge_index.shape[1]

    def edge_texts(self) -> List[str]:
        """Length-E list built by indexing the R strings (duplicates are the same object)."""
        rt = self.relation_texts
        return [rt[i] for i in self.rel_ids.tolist()]


def relation_names(R: int) -> List[str]:
    return [f"relation_{i:04d}" for i in range(R)]


def _zipf_cdf(n: int, alpha: float) -> np.ndarray:
    w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), alpha)
    c = np.cumsum(w)
    return c / c[-1]


def make_graph_arrays(N: int, E: int, R: int, seed: int, kind: str = "uniform",
                      zipf_alpha: float = 1.0) -> Tuple[np.ndarray, np.ndarray]:
    """(edge_index [2,E] int64, rel_ids [E] int64) for a uniform or power-law KG."""
    src = randint(seed, "src", E, N)
    if kind == "uniform":
        dst = randint(seed, "dst", E, N)
        rel = randint(seed, "rel", E, R)
    elif kind == "powerlaw":
        # dst ~ Zipf over a random permutation of the nodes; rel ~ Zipf over R
        cdf = _zipf_cdf(N, zipf_alpha)
        rank = np.searchsorted(cdf, uniform01(seed, "dst", E), side="right").clip(0, N - 1)
        perm = np.argsort(raw_u64(seed, "perm", N), kind="stable")
        dst = perm[rank].astype(np.int64)
        rcdf = _zipf_cdf(R, zipf_alpha)
        rel = np.searchsorted(rcdf, uniform01(seed, "rel", E), side="right").clip(0, R - 1).astype(np.int64)
    else:
        raise ValueError(f"unknown graph kind {kind!r}")
    return np.stack([src, dst]).astype(np.int64), rel.astype(np.int64)


def make_kg(N: int, E: int, R: int, F: int, seed: int, kind: str = "uniform") -> SynthKG:
    ei, rel = make_graph_arrays(N, E, R, seed, kind)
    x = normal(seed, "node_features", (N, F))
    return SynthKG(node_features=x, edge_index=ei, rel_ids=rel, relation_texts=relation_names(R))


# ---------------------------------------------------------------------------
# Parameter init with the reference's state_dict names
# ---------------------------------------------------------------------------

def weight_generator_params(prefix: str, text_dim: int, d_in: int, d_out: int, hidden_dim: int,
                            num_hidden: int, seed: int, dropout: float = 0.0,
                            init_scale: float = 0.01,
                            log_scale: Optional[float] = None) -> Dict[str, np.ndarray]:
    """Parameters of one WeightGenerator under `prefix` ("" or "weight_generators.0.").

    Distributions mimic torch defaults (U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for
    hidden Linear weight and bias) and the reference's last-layer init
    (N(0, 0.01) weight, zero bias: weight_generator.py:109-114).  Sequential
    indices step by 2, or by 3 when dropout > 0 (weight_generator.py:100-106).
    """
    p: Dict[str, np.ndarray] = {}
    step = 3 if dropout > 0.0 else 2
    for head, n_out in (("W_msg", d_in * d_out), ("W_self", d_in * d_out), ("bias", d_out)):
        prev = text_dim
        for li in range(num_hidden):
            key = f"{prefix}generators.{head}.{li * step}"
            b = 1.0 / math.sqrt(prev)
            p[key + ".weight"] = uniform(seed, key + ".weight", (hidden_dim, prev), b)
            p[key + ".bias"] = uniform(seed, key + ".bias", (hidden_dim,), b)
            prev = hidden_dim
        key = f"{prefix}generators.{head}.{num_hidden * step}"
        p[key + ".weight"] = normal(seed, key + ".weight", (n_out, prev), std=0.01)
        p[key + ".bias"] = np.zeros((n_out,), dtype=np.float32)
        ls = math.log(init_scale) if log_scale is None else log_scale
        p[f"{prefix}log_scales.{head}"] = np.full((1,), ls, dtype=np.float32)
    return p


def hypergnn_params(text_dim: int, node_feat_dim: int, hidden_dim: int, num_layers: int,
                    seed: int, char_emb_dim: int = 32, dropout: float = 0.0,
                    log_scale: Optional[float] = None,
                    randomize_ln: bool = False) -> Dict[str, np.ndarray]:
    """Full HyperGNN state_dict (reference key names, hypergnn.py:126-154) as numpy arrays."""
    p: Dict[str, np.ndarray] = {}
    p["text_encoder.char_emb.weight"] = normal(seed, "char_emb", (128, char_emb_dim))
    b = 1.0 / math.sqrt(char_emb_dim)
    p["text_encoder.proj.0.weight"] = uniform(seed, "te.w", (text_dim, char_emb_dim), b)
    p["text_encoder.proj.0.bias"] = uniform(seed, "te.b", (text_dim,), b)
    b = 1.0 / math.sqrt(node_feat_dim)
    p["input_proj.weight"] = uniform(seed, "ip.w", (hidden_dim, node_feat_dim), b)
    p["input_proj.bias"] = uniform(seed, "ip.b", (hidden_dim,), b)
    hh = max(64, 2 * text_dim)
    for l in range(num_layers):
        p.update(weight_generator_params(f"weight_generators.{l}.", text_dim, hidden_dim, hidden_dim,
                                         hh, 2, seed, dropout=dropout, log_scale=log_scale))
        if randomize_ln:
            p[f"layer_norms.{l}.weight"] = (1.0 + 0.25 * normal(seed, f"ln{l}.w", (hidden_dim,))).astype(np.float32)
            p[f"layer_norms.{l}.bias"] = (0.25 * normal(seed, f"ln{l}.b", (hidden_dim,))).astype(np.float32)
        else:
            p[f"layer_norms.{l}.weight"] = np.ones((hidden_dim,), dtype=np.float32)
            p[f"layer_norms.{l}.bias"] = np.zeros((hidden_dim,), dtype=np.float32)
    return p
