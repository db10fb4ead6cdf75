"""ms per launch of ghf_edge_outer_scaled at C3's size (one layer's weight gradients), with the slices in table order and
in the plan's launch order, and that the two give the same bits.  GHF_VARIANT picks the build."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_hypernetwork_forge_amd import _native, synth, autograd as A
from graph_hypernetwork_forge_amd.plan import build_plan
dev = torch.device("cuda:0")
N, E, R, d = int(os.environ.get("N", 1_000_000)), int(os.environ.get("E", 10_000_000)), int(os.environ.get("R", 64)), int(os.environ.get("D", 128))
reps = int(os.environ.get("REPS", 8))
ei, rel = synth.make_graph_arrays(N, E, R, 1003, os.environ.get("KIND", "uniform"))
ei_t, rel_t = torch.from_numpy(ei).to(dev), torch.from_numpy(rel).to(dev)
plan = build_plan(ei_t, rel_t, [""] * R, N, d, dev)
A.SLICE_EDGES = int(os.environ.get('SLICE', A.SLICE_EDGES))
tp = A.build_train_plan(ei_t, rel_t, plan, d, dev)
gen = torch.Generator(device=dev).manual_seed(5)
h, G = torch.randn(N, d, generator=gen, device=dev), torch.randn(N, d, generator=gen, device=dev) * 1e-3
hs, gs = _native.split_rows(h, _native.WLAYOUT_SPLIT2H), _native.split_rows(G, _native.WLAYOUT_SPLIT2H)
sc = dict(h_scales=_native.split_row_scales(hs, N, d), G_scales=_native.split_row_scales(gs, N, d))
res = {}
for name, order in (("table order", None), ("launch order", tp.slice_order)):
    f = lambda: _native.edge_outer(h, G, tp.src_by_rel, tp.dst_by_rel, tp.slice_tab, tp.slice_off, R, order=order, **sc)
    for _ in range(2):
        res[name] = f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, c in ev:
        a.record(); f(); c.record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(c) for a, c in ev]
    print(f"variant={os.environ.get('GHF_VARIANT', '-')} slices={tp.slice_tab.size(0)} {name}: {np.mean(ms):.3f} ms (min {min(ms):.3f})", flush=True)
a, b = res["table order"], res["launch order"]
print("same bits:", bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])))
