"""Host mirrors of the reference's modules (same constructors, attributes, state_dict keys and call signatures); every
forward and backward runs in libghf_hip.so."""

from .hypergnn import GraphedForward, HyperGNN, TextEncoder
from .weight_generator import WeightGenerator

__all__ = ("HyperGNN", "WeightGenerator", "TextEncoder", "GraphedForward")
