"""Public names, mirroring ``graph_hypernetwork_forge/__init__.py:24-31`` of the reference."""

__version__ = "0.2.0+mi355x.1"

from .models.weight_generator import WeightGenerator
from .models.hypergnn import HyperGNN, TextEncoder
from .data.knowledge_graph import ToyKnowledgeGraph

__all__ = ["WeightGenerator", "HyperGNN", "TextEncoder", "ToyKnowledgeGraph"]
