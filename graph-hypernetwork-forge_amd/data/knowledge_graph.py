"""ToyKnowledgeGraph — the 8-node / 11-edge / 7-relation fixture of BASELINE config 1.

Mirrors the public attributes of the reference fixture
(``graph_hypernetwork_forge/data/knowledge_graph.py:41-105``): ``node_names``,
``edge_data``, ``node_features [8, feat_dim]``, ``edge_index [2, 11]`` (int64),
``edge_texts``, ``num_nodes``, ``num_edges``, ``relation_types``.  Features are
``torch.randn`` from a generator seeded with 42, as there (:75-79); the values
the reference produced under torch 2.10 are pinned in
``tests/golden/toy_features.npz``.
"""

from __future__ import annotations

from typing import List, Optional, Tuple

import torch

_NODES = ["Alice", "Bob", "Carol", "Dave", "Eve", "Acme Corp", "London", "Python"]
_EDGES: List[Tuple[int, int, str]] = [
    (0, 1, "is spouse of"), (1, 0, "is spouse of"), (0, 2, "knows"), (1, 3, "works with"),
    (2, 3, "knows"), (3, 5, "works at"), (0, 5, "works at"), (5, 6, "located in"),
    (0, 7, "has skill"), (3, 7, "has skill"), (2, 4, "is parent of"),
]


class ToyKnowledgeGraph:
    def __init__(self, feat_dim: int = 16, node_names: Optional[List[str]] = None,
                 edge_data: Optional[List[tuple]] = None) -> None:
        self.feat_dim = feat_dim
        self.node_names = list(_NODES) if node_names is None else node_names
        self.edge_data = list(_EDGES) if edge_data is None else edge_data
        gen = torch.Generator()
        gen.manual_seed(42)
        self.node_features = torch.randn(len(self.node_names), feat_dim, generator=gen)
        self.edge_index = torch.tensor([[e[0] for e in self.edge_data], [e[1] for e in self.edge_data]],
                                       dtype=torch.long)
        self.edge_texts = [e[2] for e in self.edge_data]

    @property
    def num_nodes(self) -> int:
        return len(self.node_names)

    @property
    def num_edges(self) -> int:
        return self.edge_index.size(1)

    @property
    def relation_types(self) -> List[str]:
        return list(dict.fromkeys(self.edge_texts))

    def __repr__(self) -> str:
        return (f"ToyKnowledgeGraph(nodes={self.num_nodes}, edges={self.num_edges}, "
                f"relation_types={len(self.relation_types)})")
