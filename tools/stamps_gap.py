#!/usr/bin/env python3
"""Diagnostic (GHF_VARIANT=stamps build): how long a CU idles between two workgroups of message_bx_kernel<128> — the share of a
launch a persistent form (one workgroup per CU walking the blocks) could win back.  Per workgroup the kernel's first and last
s_memtime stamps and its CU (HW_ID, XCC_ID); per CU the workgroups in time order and the gaps between them."""
import ctypes, os, sys, collections
os.environ.setdefault("GHF_VARIANT", "stamps")
os.environ["GHF_KERNEL"] = "bx"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graph_hypernetwork_forge_amd import _build, _native, synth
from graph_hypernetwork_forge_amd.plan import build_plan
N, E, R, d = 1_000_000, 10_000_000, 64, 128
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
for _ in range(3):
    _native.message_layer_fwd(h, plan, W, None, b, plan.wlayout, g, bt, 1e-5, out, h_split=hs)
torch.cuda.synchronize()
a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); _native.message_layer_fwd(h, plan, W, None, b, plan.wlayout, g, bt, 1e-5, out, h_split=hs); c.record(); torch.cuda.synchronize()
ms = a.elapsed_time(c)
nwg = int(plan.blk_item_off[-1]) if hasattr(plan, "blk_item_off") else -(-N // plan.block_nodes)
nwg = min(nwg, 8192)
buf = np.zeros(8192 * 3, dtype=np.uint64)
fn = lib.ghf_debug_read_life_bx; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert fn(buf.ctypes.data, buf.size) == 0
life = buf.reshape(8192, 3)[:nwg]
life = life[life[:, 1] > 0]
print("first stamps: min", int(life[:, 0].min()), "p1", int(np.percentile(life[:, 0], 1)), "median", int(np.median(life[:, 0])), "max", int(life[:, 0].max()),
      "| zeros", int((life[:, 0] == 0).sum()), "| last: min", int(life[:, 1].min()), "max", int(life[:, 1].max()))
life = life[life[:, 0] > 0]
# (the XCDs' counters are not aligned with each other: times are compared on one CU only; a CU's own span ~ the launch)
per_cu = collections.defaultdict(list)
for first, last, place in life.tolist():
    per_cu[(place >> 32, (place & 0xFFFFFFFF) >> 8)].append((first, last))      # (XCC, HW_ID without wave / SIMD / pipe ids)
gaps, spans, busy, lifes = [], [], [], []
for wl in per_cu.values():
    wl.sort()
    spans.append(wl[-1][1] - wl[0][0]); busy.append(sum(l - f for f, l in wl)); lifes += [l - f for f, l in wl]
    gaps += [wl[i + 1][0] - wl[i][1] for i in range(len(wl) - 1)]
gaps, spans, busy = np.array(gaps, dtype=np.float64), np.array(spans, dtype=np.float64), np.array(busy, dtype=np.float64)
tick_us = ms * 1e3 / np.median(spans)                                           # (a CU works from the launch's start to its end, within a few %)
print(f"launch {ms:.3f} ms by events; {len(life)} workgroups on {len(per_cu)} CUs ({len(life) / len(per_cu):.1f} each); a CU's span = {np.median(spans):.0f} ticks "
      f"-> {1 / tick_us:.1f} ticks per us")
print(f"workgroup life (first to last stamp): mean {np.mean(lifes) * tick_us:.1f} us; stamped share of a CU's span: {100 * np.mean(busy / spans):.2f} %")
print(f"gap between two workgroups on a CU (last stamp of one -> first stamp of the next): mean {gaps.mean() * tick_us:.2f} us, median "
      f"{np.median(gaps) * tick_us:.2f}, p90 {np.percentile(gaps, 90) * tick_us:.2f}, max {gaps.max() * tick_us:.1f}; per CU together "
      f"{gaps.sum() / len(per_cu) * tick_us:.1f} us = {100 * gaps.sum() / spans.sum():.2f} % of the launch")
