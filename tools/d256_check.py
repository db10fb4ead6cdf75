import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
from graph_hypernetwork_forge_amd import HyperGNN, synth
dev = torch.device("cuda:0")
N, E, R, d = 100_000, 1_000_000, 64, 256
ei, rel = synth.make_graph_arrays(N, E, R, 1005)
names = synth.relation_names(R)
m = HyperGNN(text_dim=64, node_feat_dim=d, hidden_dim=d, num_layers=2).to(dev).eval().requires_grad_(False)
x = torch.randn(N, d, device=dev); eit = torch.from_numpy(ei).to(dev); relt = torch.from_numpy(rel).to(dev)
with torch.no_grad():
    m.forward_ids(x, eit, relt, names); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): m.forward_ids(x, eit, relt, names)
    torch.cuda.synchronize()
print(f"d=256: {(time.perf_counter()-t0)/3*1e3:.1f} ms per 2-layer forward at E=1M -> {E/((time.perf_counter()-t0)/3)/1e6:.1f} M edges/s")
