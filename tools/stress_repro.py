"""Bitwise reproducibility stress of one full-size layer: python tools/stress_repro.py N E R d runs [uniform|powerlaw]"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
from test_hip_parity import build_plan, _pack_weights, synth, _native, DEV   # noqa: E402

N, E, R, d, runs = [int(v) for v in sys.argv[1:6]]
kind = sys.argv[6] if len(sys.argv) > 6 else "uniform"
if kind == "uniform":
    ei, rel = synth.make_graph_arrays(N, E, R, seed=1003)
else:
    kg = synth.make_kg(N, E, R, 1, seed=1003, kind=kind)
    ei, rel = kg.edge_index, kg.rel_ids
h = torch.randn(N, d, generator=torch.Generator(device="cpu").manual_seed(1))
Wm, Ws = synth.normal(11, "Wm", (R, d, d), std=0.1), synth.normal(11, "Ws", (R, d, d), std=0.1)
b = synth.normal(11, "b", (R, d), std=0.3)
t = lambda a: torch.from_numpy(a).to(DEV)
plan = build_plan(t(ei), t(rel), [""] * R, N, d, DEV)
W = _pack_weights(plan, Wm, Ws)[0]
h_d = h.to(DEV)
hs = _native.split_rows(h_d, plan.wlayout)
args = (h_d, plan, W, None, t(b), plan.wlayout, t(np.ones(d, np.float32)), t(np.zeros(d, np.float32)), 1e-5)
ref = torch.empty_like(h_d)
_native.message_layer_fwd(*args, ref, h_split=hs)
bad, o = [], torch.empty_like(h_d)
for i in range(runs):
    _native.message_layer_fwd(*args, o, h_split=hs)
    n = int((o != ref).any(dim=1).sum().item())
    if n:
        bad.append((i, n))
print(f"variant {os.environ.get('GHF_VARIANT')} {kind} N={N} E={E} R={R} d={d} block_nodes {plan.block_nodes}: {len(bad)} of {runs} runs differ from the first", bad[:10])
