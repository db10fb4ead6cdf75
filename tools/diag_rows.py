"""Which rows of h did a wrong launch multiply?  (diagnostics, round 4)

For every wrong 8-row piece (tools/diag_attr.py's candidates): invert the relation's matrix — W_msg[r] is 64 x 64 — to get the
row x' that the consumers must have read in place of the source row (x' = x_u + err W_msg^-1), or in place of the destination row
(x'' = x_v + err W_self^-1), and look x' / x'' up among ALL rows of h (cosine; the A tiles hold rows scaled by powers of two).
A match names the row that sat in the tile; the edge lists then say which block and chunk gathers that row.

    GHF_VARIANT=b64DEFER1 python tools/diag_rows.py [launches]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
from test_hip_parity import build_plan, _pack_weights, synth, _native, DEV   # noqa: E402

N, E, R, d = 500_000, 5_000_000, 32, 64
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ei, rel = synth.make_graph_arrays(N, E, R, seed=1003)
h = torch.randn(N, d, generator=torch.Generator(device="cpu").manual_seed(1))
Wm, Ws = synth.normal(11, "Wm", (R, d, d), std=0.1), synth.normal(11, "Ws", (R, d, d), std=0.1)
b = synth.normal(11, "b", (R, d), std=0.3)
t = lambda a: torch.from_numpy(a).to(DEV)   # noqa: E731
plan = build_plan(t(ei), t(rel), [""] * R, N, d, DEV)
W = _pack_weights(plan, Wm, Ws)[0]
h_d = h.to(DEV)
hs = _native.split_rows(h_d, plan.wlayout)
outs = []
for i in range(runs):
    o = torch.empty_like(h_d)
    _native.message_layer_fwd(h_d, plan, W, None, t(b), plan.wlayout, None, None, 0.0, o, h_split=hs,
                              flags=_native.GHF_FLAG_NO_TAIL | _native.GHF_FLAG_RAW_SUM)
    torch.cuda.synchronize()
    outs.append(o)
ref = torch.stack(outs).median(dim=0).values
BN = plan.block_nodes
print("block_nodes", BN, "variant", os.environ.get("GHF_VARIANT"), flush=True)
item_tab = plan.item_tab.cpu().numpy().reshape(-1, 4)
item_off = plan.blk_item_off.cpu().numpy()
chunk_tab = plan.chunk_tab.cpu().numpy().reshape(-1, 2)
skey = plan.sorted_key.cpu().numpy().astype(np.int64)
ssrc = (plan.sorted_src.cpu().numpy().astype(np.int64)) & 0x0FFFFFFF
X = h.numpy().astype(np.float64)
Xn = h_d.double() / h_d.double().norm(dim=1, keepdim=True)
Wmi = np.linalg.inv(Wm.astype(np.float64))
Wsi = np.linalg.inv(Ws.astype(np.float64))
RPP = 8
# where a row of h is gathered as a SOURCE / as a DESTINATION: position in the sorted edge arrays -> (block, chunk)
blk_of_edge = np.zeros(plan.E, dtype=np.int64)
nb = len(item_off) - 1


def chunk_rows(ci, blk):
    e0, w1 = chunk_tab[ci]
    r, nrow = int(w1) >> 8, int(w1) & 127
    u = ssrc[e0:e0 + nrow]
    v = int(blk) * BN + (skey[e0:e0 + nrow] - (int(blk) * plan.R + r) * BN)
    return r, nrow, u, v, int(e0)


def lookup(x):
    """(row of h, cosine, scale) nearest to x."""
    xv = torch.from_numpy(x).to(DEV)
    c = Xn @ (xv / xv.norm())
    m = int(c.abs().argmax().item())
    return m, float(c[m].item()), float(np.linalg.norm(x) / np.linalg.norm(X[m]))


def where(m, near_blk):
    """how row m is used by blocks near `near_blk` in launch order: as a source of which (block, relation), or as a destination"""
    s = np.nonzero(ssrc == m)[0]
    key = skey[s]
    blks = key // (plan.R * BN)
    rels = (key % (plan.R * BN)) // BN
    src_use = sorted(zip((blks - near_blk).tolist(), rels.tolist()), key=lambda z: abs(z[0]))[:3]
    return f"destination block {m // BN} (this block {near_blk:+d}: {m // BN - near_blk:+d}); source of (block offset, relation) {src_use}"


shown = 0
for i in range(runs):
    diff = (outs[i] - ref).double()
    rows = (diff != 0).any(dim=1).nonzero().flatten().cpu().numpy()
    print(f"run {i}: {rows.size} rows off the majority", flush=True)
    dnp = {int(v): diff[int(v)].cpu().numpy() for v in rows}
    for blk in np.unique(rows // BN):
        bad = set(int(v) for v in rows[rows // BN == blk])
        for it in range(item_off[blk], item_off[blk + 1]):
            _, c0, c1, slot = item_tab[it]
            for ci in range(c0, c1):
                r, nrow, u, v, e0 = chunk_rows(ci, blk)
                for p in range((nrow + RPP - 1) // RPP):
                    vp = [int(x) for x in v[RPP * p: RPP * p + RPP]]
                    if not set(vp) <= bad or len(bad) > 40 or shown >= 30:
                        continue
                    shown += 1
                    print(f"   block {blk} chunk {ci - c0} of {c1 - c0} (item slot {slot}, r={r}, rows={nrow}) piece {p}: nodes {[x % BN for x in vp]}", flush=True)
                    for j, n in enumerate(vp):
                        if vp.count(n) != 1 or sum(1 for x in v if int(x) == n) != 1:
                            continue                                      # (a node with one row in this chunk: its error is that row's)
                        row = RPP * p + j
                        xs = X[u[row]] + dnp[n] @ Wmi[r]
                        xd = X[n] + dnp[n] @ Wsi[r]
                        ms, cs, ss = lookup(xs)
                        md, cd, sd = lookup(xd)
                        print(f"      row {row} (src {u[row]}, dst node {n % BN}): as a SOURCE row it was h[{ms}] (cos {cs:+.4f}, scale {ss:.3f});"
                              f" as a DESTINATION row h[{md}] (cos {cd:+.4f}, scale {sd:.3f})", flush=True)
                        if abs(cs) > 0.99:
                            print(f"         h[{ms}]: {where(ms, int(blk))}", flush=True)
                        if abs(cd) > 0.99:
                            print(f"         h[{md}]: {where(md, int(blk))}", flush=True)
