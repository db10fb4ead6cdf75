#!/usr/bin/env python3
"""Diagnostic: message-layer time on a power-law KG of C3 size (hubs concentrate chunks in a few destination blocks)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graph_hypernetwork_forge_amd import _native, synth
from graph_hypernetwork_forge_amd.plan import build_plan

N, E, R, d = 1_000_000, 10_000_000, 64, 128
dev = torch.device("cuda:0")
for kind in ("uniform", "powerlaw"):
    ei, rel = synth.make_graph_arrays(N, E, R, 1003, kind)
    deg = np.bincount(ei[1], minlength=N)
    plan = build_plan(torch.from_numpy(ei).to(dev), torch.from_numpy(rel).to(dev), [""] * R, N, d, dev)
    boff = plan.blk_chunk_off.cpu().numpy()
    per_blk = np.diff(boff)
    h = torch.randn(N, d, device=dev); W = torch.randn(_native.load().ghf_weights_bytes(R, d, d, plan.wlayout) // 4, device=dev) * 0.05
    if plan.wlayout in _native.SPLIT_LAYOUTS:
        W = (W.view(torch.int32) & 0x3FFF3FFF).view(torch.float32)
    b = torch.randn(R, d, device=dev); g = torch.ones(d, device=dev); bt = torch.zeros(d, device=dev)
    out = torch.empty_like(h)
    hs = _native.split_rows(h, plan.wlayout) if plan.wlayout in _native.SPLIT_LAYOUTS else None
    for _ in range(2):
        _native.message_layer_fwd(h, plan, W, None, b, plan.wlayout, g, bt, 1e-5, out, h_split=hs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        _native.message_layer_fwd(h, plan, W, None, b, plan.wlayout, g, bt, 1e-5, out, h_split=hs)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{kind:9s}: max in-degree {deg.max():8d}; chunks per block mean {per_blk.mean():7.1f} max {per_blk.max():7d}; "
          f"items {int(plan.item_off_host[-1])} (blocks {len(per_blk)}), scratch slots {plan.n_slots}; message layer {ms:8.2f} ms")
