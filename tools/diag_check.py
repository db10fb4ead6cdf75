"""GHF_VARIANT=..._bxcheck build: run a full-size hidden-64 layer a few times and print what the in-kernel tile checks recorded."""
import sys, os, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
from test_hip_parity import build_plan, _pack_weights, synth, _native, DEV   # noqa: E402

N, E, R, d = 500_000, 5_000_000, 32, 64
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ei, rel = synth.make_graph_arrays(N, E, R, seed=1003)
h = torch.randn(N, d, generator=torch.Generator(device="cpu").manual_seed(1))
Wm, Ws = synth.normal(11, "Wm", (R, d, d), std=0.1), synth.normal(11, "Ws", (R, d, d), std=0.1)
b = synth.normal(11, "b", (R, d), std=0.3)
t = lambda a: torch.from_numpy(a).to(DEV)
plan = build_plan(t(ei), t(rel), [""] * R, N, d, DEV)
W = _pack_weights(plan, Wm, Ws)[0]
h_d = h.to(DEV)
hs = _native.split_rows(h_d, plan.wlayout)
lib = _native.load()
fn = lib.ghf_debug_read_check_bx; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
buf = np.zeros(8 + 8 * 8192, dtype=np.int32)
print("block_nodes", plan.block_nodes, "variant", os.environ.get("GHF_VARIANT"))
names = {0: "P0 before phase 0", 1: "P0 after phase 0", 2: "P1 before phase 1", 3: "P1 after phase 1"}
outs = []
for i in range(runs):
    o = torch.empty_like(h_d)
    _native.message_layer_fwd(h_d, plan, W, None, t(b), plan.wlayout, None, None, 0.0, o, h_split=hs, flags=_native.GHF_FLAG_RAW_SUM)
    torch.cuda.synchronize()
    outs.append(o)
    assert fn(buf.ctypes.data, buf.size, 1) == 0
    n = int(buf[0])
    rec = buf[8:8 + 8 * min(n, 8192)].reshape(-1, 8)
    barrec = rec[(rec[:, 3] >> 4) == 12]
    gaveup = rec[(rec[:, 3] >> 4) == 14]
    print(f"run {i}: {len(gaveup)} flag waits gave up")
    sites = {0: "helper: fold flags", 1: "helper: staging flags", 10: "consumer: landed flags", 9: "consumer: staging hand-shake", 11: "consumer: last-chunk hand-shake"}
    for blk, k, w_, code, f0, f1, f2, f3 in gaveup.tolist()[:40]:
        print(f"      gave up: blk {blk} wave {w_} at {sites.get(code & 15, code & 15)} waiting for {k}: flags {f0} {f1} {f2} {f3}")
    canary = rec[(rec[:, 3] >> 4) == 13]
    print(f"run {i}: {len(canary)} canary words of the scratch KiB changed")
    for blk, nch, idx, code, got, want, _, _ in canary.tolist()[:40]:
        print(f"      canary: blk {blk} ({nch} chunks) word {idx}: {got & 0xffffffff:08x} (was {want & 0xffffffff:08x})")
    idrec = rec[((rec[:, 3] >> 4) >= 8) & ((rec[:, 3] >> 4) < 12)]
    rec = rec[(rec[:, 3] >> 4) < 8]
    print(f"run {i}: {len(barrec)} waves were past a chunk barrier before another wave had reached it")
    for blk, k, w_, code, who, seen, nch, _ in barrec.tolist()[:24]:
        print(f"      barrier: blk {blk} chunk {k} of {nch}: wave {w_} is past it, wave {who} had only announced chunk {seen - 1}")
    print(f"run {i}: {len(rec)} mismatching (row, plane) tile checks, {len(idrec)} ids wrong at DMA issue")
    for blk, k, r, code, got, want, hw, nrow in idrec.tolist()[:24]:
        print(f"      ids at issue: blk {blk} chunk {k} ({'src' if (code >> 2) & 3 == 2 else 'dst'}) row {r} of {nrow}, helper {hw}: id {got} want {want}")
    n = len(rec)
    if n and i >= 0:
        hsw = hs.view(torch.int32).cpu().numpy()[: N * d].reshape(N, d)      # per node: [hi: d/2 words | lo: d/2 words]
        item_tab = plan.item_tab.cpu().numpy().reshape(-1, 4); item_off = plan.blk_item_off.cpu().numpy()
        chunk_tab = plan.chunk_tab.cpu().numpy().reshape(-1, 2); ssrc = plan.sorted_src.cpu().numpy(); skey = plan.sorted_key.cpu().numpy().astype(np.int64)
        BN = plan.block_nodes
        def word(blk, kk, r, which, pl):
            it = item_off[blk]; c0, c1 = item_tab[it][1], item_tab[it][2]
            if not (0 <= kk < c1 - c0): return None
            e0, w1 = chunk_tab[c0 + kk]; rr, nrow = w1 >> 8, w1 & 127
            if r >= nrow: return None
            node = (ssrc[e0 + r] & ((1 << 28) - 1)) if which == 2 else blk * BN + int(skey[e0 + r] - (blk * plan.R + rr) * BN)
            key = (r >> 1) & 7
            return int(hsw[node, pl * (d // 2) + key * 4])
        shown = 0
        for blk, k, r, code, mid, wid, lv, gv in rec.tolist():
            which, pl = (code >> 2) & 3, code & 3
            cands = {f"chunk{dk:+d}/{'src' if w2 == 2 else 'dst'}/pl{p2}": word(blk, k + dk, r, w2, p2) for dk in (-4, -2, -1, 0, 1, 2) for w2 in (2, 3) for p2 in (0, 1)}
            hit = [kname for kname, v in cands.items() if v is not None and v == lv]
            if shown < 12:
                print(f"      blk {blk} k {k} row {r} code {code >> 4} {'src' if which == 2 else 'dst'} pl{pl}: lds {lv & 0xffffffff:08x} want {gv & 0xffffffff:08x} -> lds word equals: {hit}")
            shown += 1
    seen = {}
    for blk, k, r, code, mid, wid, lv, gv in rec.tolist():
        key = (blk, k, code >> 4, (code >> 2) & 3, code & 3)
        seen.setdefault(key, []).append((r, mid == wid))
    for (blk, k, c, which, pl), rows in sorted(seen.items())[:60]:
        rr = sorted(x[0] for x in rows)
        ids_ok = all(x[1] for x in rows)
        print(f"   block {blk} chunk {k}: {names[c]}, {'src' if which == 2 else 'dst'} ids {'ok' if ids_ok else 'WRONG'}, plane {pl}, rows {rr}")
ref = torch.stack(outs).median(dim=0).values
for i in range(runs):
    rows = ((outs[i] - ref) != 0).any(dim=1).nonzero().flatten().cpu().numpy()
    blks = np.unique(rows // plan.block_nodes)
    print(f"run {i}: {rows.size} output rows off the majority, blocks {blks[:12]}")
