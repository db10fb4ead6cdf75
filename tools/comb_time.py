import os, sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from graph_hypernetwork_forge_amd import _native, synth
from graph_hypernetwork_forge_amd.plan import build_plan
dev = torch.device("cuda:0")
N, E, R, d = 1_000_000, 10_000_000, 64, 128
ei, rel = synth.make_graph_arrays(N, E, R, 1003)
plan = build_plan(torch.from_numpy(ei).to(dev), torch.from_numpy(rel).to(dev), [""] * R, N, d, dev)
h = torch.randn(N, d, device=dev); W = torch.randn(_native.load().ghf_weights_bytes(R, d, d, plan.wlayout) // 4, device=dev) * 0.05
W = (W.view(torch.int32) & 0x3FFF3FFF).view(torch.float32)
b = torch.randn(R, d, device=dev); g, bt = torch.ones(d, device=dev), torch.zeros(d, device=dev)
out = torch.empty_like(h); hs = _native.split_rows(h, plan.wlayout); hso = torch.empty_like(hs)
for _ in range(3):
    _native.message_layer_fwd(h, plan, W, None, b, plan.wlayout, g, bt, 1e-5, out, h_split=hs, h_split_out=hso)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, c in ev:
    a.record(); _native.message_layer_fwd(h, plan, W, None, b, plan.wlayout, g, bt, 1e-5, out, h_split=hs, h_split_out=hso); c.record()
torch.cuda.synchronize()
print("GHF_COMB_Y", os.environ.get("GHF_COMB_Y"), "layer ms", np.mean([a.elapsed_time(c) for a, c in ev]))
