"""Small dense cases of the block kernels (D=128 default; D=64) against a float64 evaluation, with WHERE the wrong rows sit (position of
their edges in the chunk, relation, destination).  GHF_KERNEL picks the kernel (bx / pp)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_hypernetwork_forge_amd import _native
from graph_hypernetwork_forge_amd.plan import build_plan
dev = torch.device("cuda:0")
d = int(os.environ.get("D", "128"))
rng = np.random.default_rng(3)
B, C = (384, 76) if d == 128 else (256, 112)          # block nodes, rows per chunk
CASES = [  # N, R, E, description
    (B, 1, 3, "1 chunk, 3 rows"), (B, 1, 16, "1 chunk, 16 rows"), (B, 1, 20, "1 chunk, 20 rows"),
    (B, 1, 40, "1 chunk, 40 rows"), (B, 1, C, "1 full chunk"), (B, 1, C + 1, "2 chunks full+1"),
    (B, 2, 60, "2 chunks ~30"), (B, 8, 400, "8 relations x ~50"), (2 * B + 32, 4, 700, "3 blocks"), (3 * B, 3, 3000, "long blocks"),
]
for N, R, E, what in CASES:
    src = rng.integers(0, N, E); dst = rng.integers(0, N, E); rel = rng.integers(0, R, E)
    if os.environ.get("DISTINCT"):
        dst = rng.permutation(N)[:E] if E <= N else dst
    h = rng.standard_normal((N, d)).astype(np.float32)
    Wm = (rng.standard_normal((R, d, d)) * 0.1).astype(np.float32); Ws = (rng.standard_normal((R, d, d)) * 0.1).astype(np.float32)
    b = rng.standard_normal((R, d)).astype(np.float32)
    ei = np.stack([src, dst]).astype(np.int64)
    t = lambda a: torch.from_numpy(a).to(dev)
    plan = build_plan(t(ei), t(rel.astype(np.int64)), [""] * R, N, d, dev)
    W = _native.weights_pack(t(Wm), t(Ws), False, R, d, plan.wlayout)
    out = torch.full((N, d), float("nan"), device=dev)
    _native.message_layer_fwd(t(h), plan, W, None, t(b), plan.wlayout, None, None, 1e-5, out, flags=_native.GHF_FLAG_NO_TAIL)
    out2 = torch.full((N, d), float("nan"), device=dev)
    _native.message_layer_fwd(t(h), plan, W, None, t(b), plan.wlayout, None, None, 1e-5, out2, flags=_native.GHF_FLAG_NO_TAIL)
    got = out.cpu().numpy().astype(np.float64)
    ref = np.zeros((N, d)); cnt = np.zeros(N)
    h64 = h.astype(np.float64)
    for e in range(E):
        ref[dst[e]] += h64[src[e]] @ Wm[rel[e]].astype(np.float64) + b[rel[e]] + h64[dst[e]] @ Ws[rel[e]].astype(np.float64)
        cnt[dst[e]] += 1
    ref /= np.maximum(cnt, 1)[:, None]
    err = np.abs(got - ref); err[~np.isfinite(err)] = 1e30
    bad_rows = np.nonzero(err.max(1) > 1e-3)[0]
    print(f"[{what}] N={N} R={R} E={E} BN={plan.block_nodes}: max err {err.max():.3e}, bad rows {len(bad_rows)} of {int((cnt>0).sum())} with edges "
          f"({int(((cnt==0) & (err.max(1)>1e-3)).sum())} bad without edges), reproducible={bool(torch.equal(out, out2) or (torch.isnan(out) == torch.isnan(out2)).all() and torch.equal(torch.nan_to_num(out), torch.nan_to_num(out2)))}")
    if len(bad_rows):
        key = plan.sorted_key.cpu().numpy().view(np.uint32).astype(np.int64)[:E]
        bn = plan.block_nodes
        blk, rem = key // (R * bn), key % (R * bn)
        sdst = blk * bn + rem % bn
        pos_of = {}
        ctab = plan.chunk_tab.cpu().numpy()
        nch = int(plan.blk_chunk_off[-1])
        for c in range(nch):
            e0, w1 = int(ctab[2 * c]), int(ctab[2 * c + 1])
            for i in range(w1 & 127):
                pos_of.setdefault(int(sdst[e0 + i]), []).append((c, i))
        for v in bad_rows[:10]:
            cols = np.nonzero(err[v] > 1e-3)[0]
            print(f"   dst {v} (local {v % bn}): edges at (chunk,row) {pos_of.get(int(v))}; bad cols {len(cols)} first {cols[:8].tolist()}; got {got[v, cols[0]]:.4f} ref {ref[v, cols[0]]:.4f}")
        good = [v for v in np.nonzero(cnt > 0)[0] if v not in set(bad_rows.tolist())][:6]
        print("   good rows:", [(int(v), pos_of.get(int(v))) for v in good])
