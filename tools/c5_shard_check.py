#!/usr/bin/env python3
"""One GPU's share of BASELINE config 5 (power-law KG 4M nodes / 64M edges / 256 relations, hidden 256, 4 layers on 8 GPUs):
all 4M rows of h resident, 8M in-edges owned, the relation-stationary layer of csrc/message_rs.hip.  Prints ms per layer
pass and per forward (no exchange: one rank's compute).   python tools/c5_shard_check.py [--edges 8000000]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graph_hypernetwork_forge_amd import HyperGNN, _native, synth
from graph_hypernetwork_forge_amd.plan import build_plan, build_rs

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=4_000_000)
ap.add_argument("--edges", type=int, default=8_000_000)
ap.add_argument("--relations", type=int, default=256)
ap.add_argument("--hub-rows", type=int, default=0, help="override plan.RS_HUB_ROWS")
args = ap.parse_args()
if args.hub_rows:
    from graph_hypernetwork_forge_amd import plan as _plan_mod
    _plan_mod.RS_HUB_ROWS = args.hub_rows
N, E, R, d, L = args.nodes, args.edges, args.relations, 256, 4
dev = torch.device("cuda:0")
ei, rel = synth.make_graph_arrays(N, E, R, 1005, "powerlaw")
plan = build_plan(torch.from_numpy(ei).to(dev), torch.from_numpy(rel).to(dev), [""] * R, N, d, dev)
t0 = time.time(); rs = build_rs(plan); torch.cuda.synchronize(); t_rs = time.time() - t0
h = torch.randn(N, d, device=dev); out = torch.empty_like(h)
Wm = torch.randn(R, d, d, device=dev) * 0.05; Ws = torch.randn(R, d, d, device=dev) * 0.05
b = torch.randn(R, d, device=dev); g = torch.ones(d, device=dev); bt = torch.zeros(d, device=dev)
Y = rs.scratch(E, d, dev)
def layer():
    _native.edge_transform_fwd(h, rs, Wm, Ws, b, Y)
    _native.segment_tail_fwd(Y, rs, h, g, bt, 1e-5, out)
for _ in range(2): layer()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
ev[0].record(); _native.edge_transform_fwd(h, rs, Wm, Ws, b, Y); ev[1].record(); _native.segment_tail_fwd(Y, rs, h, g, bt, 1e-5, out); ev[2].record()
torch.cuda.synchronize()
p1, p2 = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
flops = 4.0 * E * d * d
print(f"N={N} E={E} R={R} d={d}: tiles {rs.slice_tab.size(0)}, rs plan {t_rs:.2f} s; pass 1 {p1:.2f} ms ({flops / p1 / 1e9:.1f} TFLOP/s fp32), "
      f"pass 2 {p2:.2f} ms; layer {p1 + p2:.2f} ms -> {L}-layer forward ~{L * (p1 + p2):.1f} ms = {E / (L * (p1 + p2)) / 1e3:.1f} M edges/s per GPU; "
      f"peak HBM {torch.cuda.max_memory_allocated() / 1e9:.1f} GB")
