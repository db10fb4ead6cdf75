import sys, time, torch
sys.path.insert(0, '/root/repo')
from graph_hypernetwork_forge_amd import HyperGNN, ToyKnowledgeGraph
dev = torch.device("cuda:0")
kg = ToyKnowledgeGraph(feat_dim=16)
torch.manual_seed(0)
m = HyperGNN(text_dim=64, node_feat_dim=16, hidden_dim=32, num_layers=2).to(dev).eval().requires_grad_(False)
x, ei = kg.node_features.to(dev), kg.edge_index.to(dev)
with torch.no_grad():
    for _ in range(5): m(x, ei, kg.edge_texts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): m(x, ei, kg.edge_texts)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 200
g = m.graphed(x, ei, kg.edge_texts)
for _ in range(5): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): g.replay()
torch.cuda.synchronize(); rep = (time.perf_counter() - t0) / 200
print(f"toy KG (C1): eager forward {eager*1e6:.0f} us, HIP-graph replay {rep*1e6:.0f} us")
