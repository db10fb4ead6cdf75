"""Raw sums of one full-size hidden-64 layer, several runs: where a run departs from the majority, and how (diagnostics)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
from test_hip_parity import build_plan, _pack_weights, synth, _native, DEV   # noqa: E402

N, E, R, d = 500_000, 5_000_000, 32, 64
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 7
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # more layer flags (4: ZERO_SRC, 8: ZERO_DST)
ei, rel = synth.make_graph_arrays(N, E, R, seed=1003)
h = torch.randn(N, d, generator=torch.Generator(device="cpu").manual_seed(1))
Wm, Ws = synth.normal(11, "Wm", (R, d, d), std=0.1), synth.normal(11, "Ws", (R, d, d), std=0.1)
b = synth.normal(11, "b", (R, d), std=0.3)
t = lambda a: torch.from_numpy(a).to(DEV)
plan = build_plan(t(ei), t(rel), [""] * R, N, d, DEV)
W = _pack_weights(plan, Wm, Ws)[0]
h_d = h.to(DEV)
hs = _native.split_rows(h_d, plan.wlayout)
outs = []
for i in range(runs):
    o = torch.empty_like(h_d)
    _native.message_layer_fwd(h_d, plan, W, None, t(b), plan.wlayout, None, None, 0.0, o, h_split=hs, flags=_native.GHF_FLAG_RAW_SUM | extra)
    torch.cuda.synchronize()
    outs.append(o)
ref = torch.stack(outs).median(dim=0).values
BN = plan.block_nodes
print("block_nodes", BN, "variant", os.environ.get("GHF_VARIANT"), "extra flags", extra)
dst, src = ei[1], ei[0]
item_tab = plan.item_tab.cpu().numpy().reshape(-1, 4)
item_off = plan.blk_item_off.cpu().numpy()
chunk_tab = plan.chunk_tab.cpu().numpy().reshape(-1, 2)
skey = plan.sorted_key.cpu().numpy().astype(np.int64)
shown = 0
for i in range(runs):
    diff = outs[i] - ref
    rows = (diff != 0).any(dim=1).nonzero().flatten().cpu().numpy()
    print(f"run {i}: {rows.size} rows off the majority")
    if not rows.size:
        continue
    for blk in np.unique(rows // BN):
        rb = rows[rows // BN == blk]
        dsub = diff[torch.from_numpy(rb).to(DEV)].cpu().numpy()
        ncols = (dsub != 0).sum(axis=1)
        tot = np.abs(dsub.sum(axis=0)).max()
        e_in = np.isin(dst, rb)
        rels = [np.unique(rel[e_in & (dst == v)]) for v in rb]
        common = set(rels[0].tolist())
        for s in rels[1:]:
            common &= set(s.tolist())
        # does a bad row's error equal a whole number of its in-edges of one relation?  indeg per relation for the bad rows
        print(f"   block {blk}: {rb.size} rows, local {(rb % BN)[:12]}..., cols/row min {ncols.min()} max {ncols.max()}, "
              f"|row err| {np.abs(dsub).max(axis=1)[:6].round(3)}, |sum of errors| {tot:.3f}, relations common to all bad rows: {sorted(common)[:8]}")
        # the block's items and, per item, the chunks; which chunk holds the bad rows, and at which row positions
        for it in range(item_off[blk], item_off[blk + 1]):
            _, c0, c1, slot = item_tab[it]
            for ci in range(c0, c1):
                e0, w1 = chunk_tab[ci]
                r, nrow = w1 >> 8, w1 & 127
                loc = skey[e0:e0 + nrow] - (int(blk) * plan.R + r) * BN          # local destination of the chunk's rows
                pos = np.nonzero(np.isin(loc, rb % BN))[0]
                if pos.size and pos.size >= min(rb.size, 4) and set((rb % BN).tolist()) <= set(loc.tolist()):
                    print(f"      item {it - item_off[blk]}/{item_off[blk + 1] - item_off[blk]} slot {slot} chunk {ci - c0} of {c1 - c0}: r={r} rows={nrow}, "
                          f"bad rows at positions {pos[:20]}")
        shown += 1
        if shown > 40:
            sys.exit(0)
