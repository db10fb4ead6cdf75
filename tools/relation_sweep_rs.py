"""Where the relation-stationary layer overtakes the destination-block kernel at d = 128: the same 1 M-node / 10 M-edge graph with
R relations through message_bx and through run_rows/edge_transform_h/segment_tail (ms per layer)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_hypernetwork_forge_amd import _native, synth
from graph_hypernetwork_forge_amd.plan import build_plan, build_rs
dev = torch.device("cuda:0")
N, E, d = 1_000_000, 10_000_000, 128
def timed(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, c in ev:
        a.record(); fn(); c.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(c) for a, c in ev]))
for R in [int(x) for x in os.environ.get("RS", "64,96,128,192,256").split(",")]:
    ei, rel = synth.make_graph_arrays(N, E, R, 1003)
    t = lambda a: torch.from_numpy(a).to(dev)
    h = torch.randn(N, d, device=dev)
    Wm, Ws = torch.randn(R, d, d, device=dev) * 0.05, torch.randn(R, d, d, device=dev) * 0.05
    b = torch.randn(R, d, device=dev); g = torch.ones(d, device=dev); bt = torch.zeros(d, device=dev)
    out = torch.empty_like(h)
    os.environ["GHF_KERNEL"] = "bx"
    plan = build_plan(t(ei), t(rel), [""] * R, N, d, dev)
    W = _native.weights_pack(Wm, Ws, False, R, d, plan.wlayout)
    hs = _native.split_rows(h, plan.wlayout)
    ms_bx = timed(lambda: _native.message_layer_fwd(h, plan, W, None, b, plan.wlayout, g, bt, 1e-5, out, h_split=hs))
    os.environ["GHF_KERNEL"] = "rs"
    planr = build_plan(t(ei), t(rel), [""] * R, N, d, dev, force_generic=True)
    rs = build_rs(planr)
    Y = rs.scratch(E, d, dev)
    def layer():
        _native.edge_transform_fwd(h, rs, Wm, Ws, b, Y, h_split=hs)
        _native.segment_tail_fwd(Y, rs, h, g, bt, 1e-5, out)
    ms_rs = timed(layer)
    print(f"R={R:4d} message_bx {ms_bx:.3f} ms   relation-stationary {ms_rs:.3f} ms (rows {rs.rows})", flush=True)
