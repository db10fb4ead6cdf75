#!/usr/bin/env python3
"""Race hunt for the block kernels' hand-over (flags, counted waits, deferred staging): many launches of the same layer —
uniform and power-law graphs (split hub blocks), hidden 128 and 64, the plain launch, the side output and the two
zero-half gradient passes — every result compared bit for bit with the first.  usage: tools/bx_stress.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graph_hypernetwork_forge_amd import _native, synth
from graph_hypernetwork_forge_amd.plan import build_plan
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
bad = 0
for d, N, E, R, kind in ((128, 1_000_000, 10_000_000, 64, "uniform"), (128, 200_000, 3_000_000, 16, "powerlaw"),
                         (64, 500_000, 5_000_000, 32, "uniform"), (128, 40_000, 300_000, 200, "uniform")):
    os.environ["GHF_KERNEL"] = "bx"
    ei, rel = synth.make_graph_arrays(N, E, R, 7 + d + R, kind)
    plan = build_plan(torch.from_numpy(ei).to(dev), torch.from_numpy(rel).to(dev), [""] * R, N, d, dev)
    g = torch.Generator(device=dev).manual_seed(3)
    h = torch.randn(N, d, device=dev, generator=g)
    Wm = torch.randn(R, d, d, device=dev, generator=g) * 0.1
    Ws = torch.randn(R, d, d, device=dev, generator=g) * 0.1
    b = torch.randn(R, d, device=dev, generator=g)
    gm, bt = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    hs = _native.split_rows(h, plan.wlayout)
    W = _native.weights_pack(Wm, Ws, False, R, d, plan.wlayout)
    Wz = _native.weights_pack(None, Ws, True, R, d, plan.wlayout)
    Wy = _native.weights_pack(Wm, None, True, R, d, plan.wlayout)
    modes = {"plain": dict(W=W, flags=0, tail=True), "side": dict(W=W, flags=0, tail=True, side=True),
             "zero_src": dict(W=Wz, flags=_native.GHF_FLAG_RAW_SUM | _native.GHF_FLAG_ZERO_SRC, tail=False),
             "zero_dst": dict(W=Wy, flags=_native.GHF_FLAG_RAW_SUM | _native.GHF_FLAG_ZERO_DST, tail=False)}
    for name, m in modes.items():
        first = None
        n_bad = 0
        for i in range(reps if name == "plain" else max(reps // 5, 10)):
            out = torch.empty_like(h)
            agg = torch.empty_like(h) if m.get("side") else None
            hso = torch.empty_like(hs) if m["tail"] else None
            _native.message_layer_fwd(h, plan, m["W"], None, b, plan.wlayout, gm if m["tail"] else None, bt if m["tail"] else None,
                                      1e-5, out, h_split=hs, h_split_out=hso, agg_out=agg, flags=m["flags"])
            key = (out, agg, hso)
            if first is None:
                first = key
                assert bool(torch.isfinite(out).all())
            else:
                same = all(a is None or torch.equal(a.view(torch.uint8) if a.dtype != torch.float32 else a, c.view(torch.uint8) if c.dtype != torch.float32 else c)
                           for a, c in zip(key, first))
                n_bad += 0 if same else 1
        torch.cuda.synchronize()
        bad += n_bad
        print(f"d={d} {kind} N={N} E={E} R={R} BN={plan.block_nodes} items={int(plan.item_off_host[-1])} {name}: {n_bad} launches differ from the first", flush=True)
print("FAILED" if bad else "all launches bitwise equal")
sys.exit(1 if bad else 0)
