from .knowledge_graph import ToyKnowledgeGraph

__all__ = ["ToyKnowledgeGraph"]
