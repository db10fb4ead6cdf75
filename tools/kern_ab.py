"""A/B of the d = 128 message kernels on one box: one C3-sized layer per GHF_KERNEL value (KERNELS=pp,bx), ms per launch
by HIP events, and the largest difference of each kernel's output from the first one's."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_hypernetwork_forge_amd import _native, synth
from graph_hypernetwork_forge_amd.plan import build_plan
dev = torch.device("cuda:0")
N, E, R, d = int(os.environ.get("N", 1_000_000)), int(os.environ.get("E", 10_000_000)), int(os.environ.get("R", 64)), 128
kind = os.environ.get("KIND", "uniform")
reps = int(os.environ.get("REPS", 10))
ei, rel = synth.make_graph_arrays(N, E, R, 1003, kind)
gen = torch.Generator(device=dev).manual_seed(5)
h = torch.randn(N, d, generator=gen, device=dev)
Wm = torch.randn(R, d, d, generator=gen, device=dev) * 0.05
Ws = torch.randn(R, d, d, generator=gen, device=dev) * 0.05
b = torch.randn(R, d, generator=gen, device=dev); g = torch.ones(d, device=dev); bt = torch.zeros(d, device=dev)
ref = None
for kern in os.environ.get("KERNELS", "pp,bx").split(","):
    os.environ["GHF_KERNEL"] = kern
    plan = build_plan(torch.from_numpy(ei).to(dev), torch.from_numpy(rel).to(dev), [""] * R, N, d, dev)
    W = _native.weights_pack(Wm, Ws, False, R, d, plan.wlayout)
    hs = _native.split_rows(h, plan.wlayout) if plan.wlayout in _native.SPLIT_LAYOUTS else None
    out = torch.empty_like(h)
    for _ in range(2):
        _native.message_layer_fwd(h, plan, W, None, b, plan.wlayout, g, bt, 1e-5, out, h_split=hs)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, c in ev:
        a.record(); _native.message_layer_fwd(h, plan, W, None, b, plan.wlayout, g, bt, 1e-5, out, h_split=hs); c.record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(c) for a, c in ev]
    again = torch.empty_like(h)
    _native.message_layer_fwd(h, plan, W, None, b, plan.wlayout, g, bt, 1e-5, again, h_split=hs)
    msg = f"kernel={kern:3s} BN={plan.block_nodes} CR={plan.chunk_rows} chunks={int(plan.blk_chunk_off[-1])} ms={np.mean(ms):.3f} (min {min(ms):.3f})"
    msg += f" reproducible={bool(torch.equal(out, again))} finite={bool(torch.isfinite(out).all())}"
    if ref is None:
        ref = out.clone()
    else:
        diff = (out - ref).abs()
        diff = torch.where(torch.isfinite(diff), diff, torch.full_like(diff, 1e30))
        msg += f" max|diff vs first|={float(diff.max()):.3e}"
        bad = torch.nonzero(diff.max(dim=1).values > 1e-3).flatten()
        if bad.numel():
            bn = plan.block_nodes
            loc = (bad % bn).cpu().numpy()
            msg += (f"\n   bad rows: {bad.numel()} of {N}; first {bad[:12].tolist()}; block-local first {loc[:12].tolist()}; "
                    f"bad per quarter of a block {np.bincount(loc * 4 // bn, minlength=4).tolist()}; "
                    f"bad rows with in-degree 0: {int((plan.indeg[bad] == 0).sum())}; "
                    f"cols of first bad row {torch.nonzero(diff[bad[0]] > 1e-3).flatten()[:16].tolist()} n={int((diff[bad[0]] > 1e-3).sum())}")
    print(msg, flush=True)
