"""Graph fixtures of the drop-in: the reference's toy knowledge graph (same nodes, edges and relation strings), used by
the demo-shaped tests and as BASELINE config 1."""

from . import knowledge_graph as _kg

ToyKnowledgeGraph = _kg.ToyKnowledgeGraph

__all__ = ("ToyKnowledgeGraph",)
