"""WeightGenerator — host mirror of the reference module, computing on MI355X.

Same constructor, attributes, ``state_dict`` keys and output dict as
``graph_hypernetwork_forge/models/weight_generator.py:50-143`` of the
reference; ``forward`` runs the relation-batched HIP kernels of
``csrc/weightgen.hip`` through the C ABI (``ghf_weightgen_fwd``) instead of
three ``nn.Sequential`` stacks.  The ``nn.Linear`` modules exist to hold the
parameters under the reference's names; they are never called.  When gradients
are required the same kernels run inside ``autograd.WeightGeneratorFn``, whose
backward is C-ABI calls as well.
"""

from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import _native

HEADS: Tuple[str, ...] = ("W_msg", "W_self", "bias")     # reference weight_generator.py:72-76


def check_dropout(p: float) -> None:
    """nn.Dropout / F.dropout of the reference raise ValueError outside [0, 1] (weight_generator.py:104, hypergnn.py:293)."""
    if not 0.0 <= float(p) <= 1.0:
        raise ValueError(f"dropout probability has to be between 0 and 1, but got {p}")


def draw_mask(shape, device, p: float) -> torch.Tensor:
    """Bernoulli(1 - p) / (1 - p), drawn with torch's generator as F.dropout / nn.Dropout do; p = 1 drops everything (the
    reference returns zeros there; 0 / 0 would be NaN)."""
    if p >= 1.0:
        return torch.zeros(shape, dtype=torch.float32, device=device)
    return (torch.rand(shape, device=device) >= p).to(torch.float32) / (1.0 - p)


def wants_grad(module: nn.Module, *tensors: torch.Tensor) -> bool:
    """Checks that the inputs live on the HIP device; True when autograd has to record this call."""
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                f"{type(module).__name__} computes on an MI355X HIP device only (input is on {t.device}); "
                "this package has no CPU or eager-PyTorch fallback")
    return torch.is_grad_enabled() and (any(t.requires_grad for t in tensors) or
                                        any(p.requires_grad for p in module.parameters()))


def require_inference(module: nn.Module, *tensors: torch.Tensor, what: str = "") -> None:
    """For the few entry points that have no backward on the HIP path."""
    if wants_grad(module, *tensors):
        raise NotImplementedError(
            f"{type(module).__name__}{what}: no backward on the HIP path; call it under torch.no_grad() "
            "(or after .requires_grad_(False)).")


class WeightGenerator(nn.Module):
    """Generate ``(W_msg, W_self, bias)`` of one GNN layer from relation-text embeddings.

    Args mirror the reference (weight_generator.py:50-59): ``text_dim, d_in,
    d_out, hidden_dim=128, num_hidden=2, dropout=0.0, init_scale=0.01``.
    """

    def __init__(self, text_dim: int, d_in: int, d_out: int, hidden_dim: int = 128, num_hidden: int = 2,
                 dropout: float = 0.0, init_scale: float = 0.01) -> None:
        super().__init__()
        if text_dim <= 0 or d_in <= 0 or d_out <= 0:             # reference :62-63
            raise ValueError("text_dim, d_in, d_out must all be positive integers")
        check_dropout(dropout)                                   # (the reference's nn.Dropout(p) raises the same way, :104)
        self.text_dim, self.d_in, self.d_out = text_dim, d_in, d_out
        self.hidden_dim, self.num_hidden, self.dropout = hidden_dim, num_hidden, dropout
        self.init_scale = init_scale
        self._weight_specs: List[Tuple[str, Tuple[int, ...]]] = [
            ("W_msg", (d_in, d_out)), ("W_self", (d_in, d_out)), ("bias", (d_out,))]

        # Parameter containers laid out so that state_dict keys equal the reference's
        # (generators.<head>.<idx>.{weight,bias}; idx steps by 3 when Dropout modules sit between).
        self.generators = nn.ModuleDict()
        for name, shape in self._weight_specs:
            mods: List[nn.Module] = []
            width = text_dim
            for _ in range(num_hidden):
                mods += [nn.Linear(width, hidden_dim), nn.ReLU()]
                if dropout > 0.0:
                    mods.append(nn.Dropout(dropout))
                width = hidden_dim
            last = nn.Linear(width, math.prod(shape))
            nn.init.zeros_(last.bias)                              # reference :109-114
            nn.init.normal_(last.weight, std=0.01)
            mods.append(last)
            self.generators[name] = nn.Sequential(*mods)
        self.log_scales = nn.ParameterDict({
            name: nn.Parameter(torch.full((1,), math.log(init_scale))) for name, _ in self._weight_specs})

    # -- parameter packing for the C ABI ---------------------------------------------
    def _linears(self, head: str) -> List[nn.Linear]:
        return [m for m in self.generators[head] if isinstance(m, nn.Linear)]

    def _head_params(self) -> List[torch.Tensor]:
        flat: List[torch.Tensor] = []
        for head in HEADS:
            for lin in self._linears(head):
                flat += [lin.weight.detach(), lin.bias.detach()]
        return flat

    def _head_parameters(self) -> List[torch.Tensor]:
        return [p for head in HEADS for lin in self._linears(head) for p in (lin.weight, lin.bias)]

    def _dropping(self) -> bool:
        return self.training and self.dropout > 0.0 and self.num_hidden > 0

    def _draw_mask(self, shape, device) -> torch.Tensor:
        """Dropout masks scaled by 1/(1-p), drawn with torch's generator as the reference's nn.Dropout modules do."""
        return draw_mask(shape, device, self.dropout)

    def generate_with_grad(self, text_emb: torch.Tensor):
        """Natural-layout (W_msg, W_self, bias) recorded by autograd (backward through the C ABI); in training mode with
        dropout > 0 the hidden activations are masked (reference :96-107: Linear -> ReLU -> Dropout)."""
        from ..autograd import WeightGeneratorFn
        dims = (self.text_dim, self.hidden_dim, self.num_hidden, self.d_in, self.d_out)
        if self._dropping():
            masks = self._draw_mask((3, self.num_hidden, text_emb.size(0), self.hidden_dim), text_emb.device)
            # (p = 1: every unit is dropped and its gradient is zero whatever the factor: 0, not log(1/0))
            log_keep = torch.full((1,), -math.log(1.0 - self.dropout) if self.dropout < 1.0 else 0.0, dtype=torch.float32,
                                  device=text_emb.device)
            dims = dims + (masks, log_keep)
        return WeightGeneratorFn.apply(dims, text_emb, *(self.log_scales[h] for h in HEADS), *self._head_parameters())

    def _log_scale_vector(self) -> List[torch.Tensor]:
        """The three 1-element log-scale parameters, read in place by ghf_weightgen_fwd (no concatenation kernel)."""
        return [self.log_scales[h].detach() for h in HEADS]

    def generate(self, text_emb: torch.Tensor, layout: int = _native.WLAYOUT_NATURAL):
        """[R,T] embeddings -> (W_msg | Wfrag, W_self | None, bias) device tensors in `layout`."""
        return _native.weightgen_fwd(text_emb, self._head_params(), self._log_scale_vector(), self.text_dim,
                                     self.hidden_dim, self.num_hidden, self.d_in, self.d_out, layout)

    # -- public API (reference :120-143) -----------------------------------------------
    def forward(self, text_emb: torch.Tensor) -> Dict[str, torch.Tensor]:
        single = text_emb.dim() == 1
        if text_emb.size(-1) != self.text_dim:
            raise ValueError(f"text_emb has last dim {text_emb.size(-1)}, expected text_dim={self.text_dim}")
        grad = wants_grad(self, text_emb) or self._dropping()
        x = text_emb.unsqueeze(0) if single else text_emb
        W_msg, W_self, bias = self.generate_with_grad(x.float()) if grad else self.generate(x.float())
        out = {"W_msg": W_msg, "W_self": W_self, "bias": bias}
        return {k: v.squeeze(0) for k, v in out.items()} if single else out
