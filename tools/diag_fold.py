"""Which ROWS went wrong in a wrong launch of the hidden-64 block kernel, as a combination of the chunk's staged rows (round 4).

For every block with wrong nodes: the chunk that holds them; then every wrong node's error is written as a combination
sum_i c_i Y_i of the staged rows Y_i = h_u W_msg[r] + b[r] + h_v W_self[r] of that chunk (least squares over the rows within
a few positions of the node's own rows).  A fold that lost a row shows c = -1 on it, one that added a row to the wrong node
c = +1 there, a wrong operand no clean combination at all (residual printed).

    GHF_VARIANT=b64DEFER1_bxFOLDDELAY30 python tools/diag_fold.py [launches]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
from test_hip_parity import build_plan, _pack_weights, synth, _native, DEV   # noqa: E402

N, E, R, d = 500_000, 5_000_000, 32, 64
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 7
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
NPW = BN // 4
print("block_nodes", BN, "variant", os.environ.get("GHF_VARIANT"), flush=True)
item_tab = plan.item_tab.cpu().numpy().reshape(-1, 4)
item_off = plan.blk_item_off.cpu().numpy()
chunk_tab = plan.chunk_tab.cpu().numpy().reshape(-1, 2)
skey = plan.sorted_key.cpu().numpy().astype(np.int64)
ssrc = (plan.sorted_src.cpu().numpy().astype(np.int64)) & 0x0FFFFFFF
X = h.numpy().astype(np.float64)
Wm64, Ws64, b64 = Wm.astype(np.float64), Ws.astype(np.float64), b.astype(np.float64)
refn = ref.double().cpu().numpy()


def chunk_rows(ci, blk):
    e0, w1 = chunk_tab[ci]
    r, nrow = int(w1) >> 8, int(w1) & 127
    u = ssrc[e0:e0 + nrow]
    v = int(blk) * BN + (skey[e0:e0 + nrow] - (int(blk) * plan.R + r) * BN)
    return r, nrow, u, v


shown = 0
for i in range(runs):
    diff = (outs[i] - ref).double()
    rows = (diff != 0).any(dim=1).nonzero().flatten().cpu().numpy()
    print(f"run {i}: {rows.size} rows off the majority", flush=True)
    for blk in np.unique(rows // BN):
        if shown >= 24:
            break
        bad = sorted(int(v) for v in rows[rows // BN == blk])
        err = {v: diff[v].cpu().numpy() for v in bad}
        # the chunk that holds most of the wrong nodes
        best = None
        for it in range(item_off[blk], item_off[blk + 1]):
            _, c0, c1, slot = item_tab[it]
            for ci in range(c0, c1):
                r, nrow, u, v = chunk_rows(ci, blk)
                cover = len(set(int(x) for x in v) & set(bad))
                pos = [p for p in range(nrow) if int(v[p]) in err]
                span = (max(pos) - min(pos) + 1) if pos else 999
                key = (cover, -span)
                if best is None or key > best[0]:
                    best = (key, it, c0, c1, ci, slot)
        _, it, c0, c1, ci, slot = best
        r, nrow, u, v = chunk_rows(ci, blk)
        Y = X[u] @ Wm64[r] + X[v] @ Ws64[r] + b64[r][None, :]
        loc = v - int(blk) * BN
        pos = [p for p in range(nrow) if int(v[p]) in err]
        ra = [int(np.searchsorted(loc, NPW * w)) for w in range(5)]         # first row of every helper wave (rows sorted by destination)
        shown += 1
        print(f"   block {blk}: {len(bad)} wrong nodes {[x % BN for x in bad][:14]}; chunk {ci - c0} of {c1 - c0} (item {it - item_off[blk]}, slot {slot}, r={r}, rows={nrow}) "
              f"covers {best[0][0]}; wrong-node rows at positions {pos}; helper waves' first rows {ra[:4]}", flush=True)
        lo, hi = max(0, min(pos) - 8), min(nrow, max(pos) + 9)
        A = Y[lo:hi].T                                                       # [d, rows considered]
        for n in bad[:12]:
            c, res, rk, sv = np.linalg.lstsq(A, err[n], rcond=None)
            fit = np.linalg.norm(A @ c - err[n]) / max(np.linalg.norm(err[n]), 1e-30)
            nz = [(lo + j, round(float(c[j]), 3)) for j in range(hi - lo) if abs(c[j]) > 0.02]
            own = [p for p in range(nrow) if int(v[p]) == n]
            # other readings of the same error: a whole node sum swapped in?
            alt = ""
            if fit > 0.05:
                cand = [(float(np.linalg.norm(err[n] - s * refn[m]) / np.linalg.norm(err[n])), m % BN, s)
                        for m in range(int(blk) * BN, min(int(blk) * BN + BN, N)) for s in (1.0, -1.0)]
                cand.sort()
                alt = f"; nearest whole-node sum: node {cand[0][1]} x {cand[0][2]:+.0f} (residual {cand[0][0]:.2f})"
            print(f"      node {n % BN} (wave {(n % BN) // NPW}; its rows {own}): |err| {np.linalg.norm(err[n]):.3f} = {nz}  fit residual {fit:.1e}{alt}", flush=True)
