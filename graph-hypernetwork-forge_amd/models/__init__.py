from .weight_generator import WeightGenerator
from .hypergnn import HyperGNN, TextEncoder

__all__ = ["WeightGenerator", "HyperGNN", "TextEncoder"]
