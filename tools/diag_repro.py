"""Run one full-size hidden-64 layer several times and describe where two runs differ (diagnostics for a block-kernel race).
usage: python tools/diag_repro.py [N E R d [runs]]"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
from test_hip_parity import build_plan, _pack_weights, synth, _native, DEV   # noqa: E402

N, E, R, d = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (500_000, 5_000_000, 32, 64)
runs = int(sys.argv[5]) if len(sys.argv) > 5 else 4
ei, rel = synth.make_graph_arrays(N, E, R, seed=1003)
h = torch.randn(N, d, generator=torch.Generator(device="cpu").manual_seed(1))
Wm, Ws = synth.normal(11, "Wm", (R, d, d), std=0.1), synth.normal(11, "Ws", (R, d, d), std=0.1)
b = synth.normal(11, "b", (R, d), std=0.3)
t = lambda a: torch.from_numpy(a).to(DEV)
plan = build_plan(t(ei), t(rel), [""] * R, N, d, DEV)
W = _pack_weights(plan, Wm, Ws)[0]
h_d = h.to(DEV)
args = (h_d, plan, W, None, t(b), plan.wlayout, t(np.ones(d, np.float32)), t(np.zeros(d, np.float32)), 1e-5)
outs = []
for i in range(runs):
    o = torch.empty_like(h_d)
    _native.message_layer_fwd(*args, o)
    torch.cuda.synchronize()
    outs.append(o)
BN = plan.block_nodes
print("block_nodes", BN, "pad", os.environ.get("GHF_BX_LDS_PAD"), "variant", os.environ.get("GHF_VARIANT"))
for i in range(1, runs):
    diff = (outs[i] != outs[0])
    rows = diff.any(dim=1).nonzero().flatten().cpu().numpy()
    print(f"run {i}: {rows.size} rows differ")
    if rows.size:
        blk = rows // BN
        ub, cnt = np.unique(blk, return_counts=True)
        print("  blocks:", ub[:20], "rows per block:", cnt[:20])
        loc = rows % BN
        print("  local nodes:", loc[:40])
        r0 = int(rows[0])
        cols = diff[r0].nonzero().flatten().cpu().numpy()
        print("  row", r0, "cols differing:", cols.size, cols[:16], "max abs diff", float((outs[i][r0] - outs[0][r0]).abs().max()))
        print("  indeg of differing rows:", plan.indeg[torch.from_numpy(rows[:20]).to(DEV)].cpu().numpy())
