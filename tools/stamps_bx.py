#!/usr/bin/env python3
"""Diagnostic: where message_bx.hip's waves spend their cycles (GHF_VARIANT=stamps build; shares only)."""
import ctypes, os, sys
os.environ.setdefault("GHF_VARIANT", "stamps")
os.environ["GHF_KERNEL"] = "bx"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graph_hypernetwork_forge_amd import _build, _native, synth
from graph_hypernetwork_forge_amd.plan import build_plan
N, E, R, d = (100_000, 1_000_000, 32, 64) if os.environ.get("D") == "64" else (1_000_000, 10_000_000, 64, 128)
_build.build()
lib = _native.load()
dev = torch.device("cuda:0")
ei, rel = synth.make_graph_arrays(N, E, R, 1003)
plan = build_plan(torch.from_numpy(ei).to(dev), torch.from_numpy(rel).to(dev), [""] * R, N, d, dev)
h = torch.randn(N, d, device=dev)
W = torch.randn(lib.ghf_weights_bytes(R, d, d, plan.wlayout) // 4, device=dev) * 0.05
W = (W.view(torch.int32) & 0x3FFF3FFF).view(torch.float32)
b = torch.randn(R, d, device=dev); g, bt = torch.ones(d, device=dev), torch.zeros(d, device=dev)
out = torch.empty_like(h); hs = _native.split_rows(h, plan.wlayout)
for _ in range(2):
    _native.message_layer_fwd(h, plan, W, None, b, plan.wlayout, g, bt, 1e-5, out, h_split=hs)
torch.cuda.synchronize()
nb = min(8192, -(-N // plan.block_nodes))
buf = np.zeros(8192 * 8 * 8, dtype=np.uint64)
fn = lib.ghf_debug_read_stamps_bx; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert fn(buf.ctypes.data, buf.size) == 0
st = buf.reshape(8192, 8, 8)[:nb].astype(np.float64)
for role, sl, names in (("consumers", slice(0, 4), ["barrier wait", "phase-0: unscale+rest", "phase-1: unscale+rest", "staging writes", "stage prologue", "k-step MFMAs", "epilogue + first half of the tail", "B refill issue"]),
                        ("helpers", slice(4, 8), ["barrier wait", "DMA issue", "fold (rows into registers)", "wait for the other helpers", "descriptor work + DMA landing", "wait for the staged rows (+ the epilogue's fold)", "tail: first half", "tail: second half"])):
    x = st[:, sl]
    tot = x.sum()
    print(f"{role}: mean cycles per wave {x.sum(axis=2).mean():.0f}")
    for i, n in enumerate(names):
        if n != "-":
            print(f"   {n:18s} {100 * x[:, :, i].sum() / tot:6.2f} %   per chunk {x[:, :, i].mean() / (plan.blk_chunk_off[-1].item() / nb):8.0f}")
