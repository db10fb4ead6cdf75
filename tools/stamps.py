#!/usr/bin/env python3
"""Diagnostic: where a message-kernel loop iteration spends its cycles (GHF_VARIANT=stamps build).

Usage (GPU box):  GHF_VARIANT=stamps python tools/stamps.py [c3|c2]
Prints the share of per-wave cycles in: memory wait, barrier, prefetch issue, compute, scatter, tail.
Shares only — the stamped build is slower than the product build and its run time is never quoted.
"""
import ctypes
import os
import sys

os.environ.setdefault("GHF_VARIANT", "stamps")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from graph_hypernetwork_forge_amd import _build, _native, synth
from graph_hypernetwork_forge_amd.plan import build_plan

cfgs = {"c3": (1_000_000, 10_000_000, 64, 128), "c2": (100_000, 1_000_000, 32, 64)}
N, E, R, d = cfgs[sys.argv[1] if len(sys.argv) > 1 else "c3"]
_build.build()
lib = _native.load()
dev = torch.device("cuda:0")
ei, rel = synth.make_graph_arrays(N, E, R, 1003)
plan = build_plan(torch.from_numpy(ei).to(dev), torch.from_numpy(rel).to(dev), [""] * R, N, d, dev)
h = torch.randn(N, d, device=dev)
W = torch.randn(_native.load().ghf_weights_bytes(R, d, d, plan.wlayout) // 4, device=dev) * 0.05
if plan.wlayout in _native.SPLIT_LAYOUTS:          # any finite bf16 bit patterns will do for timing
    W = (W.view(torch.int32) & 0x3FFF3FFF).view(torch.float32)
b = torch.randn(R, d, device=dev)
g, bt = torch.ones(d, device=dev), torch.zeros(d, device=dev)
out = torch.empty_like(h)
hs = _native.split_rows(h, plan.wlayout) if plan.wlayout in _native.SPLIT_LAYOUTS else None
for _ in range(2):
    _native.message_layer_fwd(h, plan, W, None, b, plan.wlayout, g, bt, 1e-5, out, h_split=hs)
torch.cuda.synchronize()
nb = min(8192, -(-N // plan.block_nodes))
buf = np.zeros(8192 * 8 * 8, dtype=np.uint64)
# message_bx.hip has its own reader (tools/stamps_bx.py); here: the exact fp32 kernel (GHF_KERNEL=pp)
if plan.wlayout in _native.SPLIT_LAYOUTS:
    raise SystemExit("the default kernels are stamped by tools/stamps_bx.py; set GHF_KERNEL=pp for this tool")
fn = lib.ghf_debug_read_stamps_pp
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert fn(buf.ctypes.data, buf.size) == 0
st = buf.reshape(8192, 8, 8)[:nb].astype(np.float64)
tot = st.sum(axis=2)
names = ["barrier wait", "mfma interval", "prep: rest of issue", "prep: scatter", "prep: mem wait", "prep: words+shuffles", "drain+tail", "prep: gather issue"]   # ping-pong kernel (message_pp.hip)
print(f"blocks={nb} mean cycles per wave = {tot.mean():.0f}")
g0, g1 = "team0", "team1"
for i, n in enumerate(names[:8]):
    print(f"  {n:24s} {100 * st[:, :, i].sum() / tot.sum():6.2f} %   mean {st[:, :, i].mean():10.0f}   {g0} {st[:, :4, i].mean():10.0f}  {g1} {st[:, 4:, i].mean():10.0f}")
