"""ghf_weightgen_bwd alone: wall time per call (launch-bound sizes) and device time per call (HIP events), at the generator
shapes of BASELINE configs 1-3.  python tools/wgb_time.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_hypernetwork_forge_amd import _native
dev = torch.device("cuda:0")
for name, R, T, Hh, nh, d in (("c1", 7, 64, 128, 2, 32), ("c2", 32, 64, 128, 2, 64), ("c3", 64, 64, 128, 2, 128)):
    g = torch.Generator(device=dev).manual_seed(1)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)       # noqa: E731
    x = rn(R, T)
    params = []
    for k in range(3):
        dims = [T] + [Hh] * nh + [d if k == 2 else d * d]
        for l in range(nh + 1):
            params += [0.1 * rn(dims[l + 1], dims[l]), 0.1 * rn(dims[l + 1])]
    ls = torch.zeros(3, device=dev)
    acts = _native.weightgen_acts(x, params, T, Hh, nh)
    outs = [rn(R, d * d), rn(R, d * d), rn(R, d)]
    grads = [rn(R, d * d), rn(R, d * d), rn(R, d)]
    f = lambda: _native.weightgen_bwd(x, params, acts, outs, grads, ls, T, Hh, nh, d, d)   # noqa: E731
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize()
    print(f"{name}: R={R} d={d}: {wall:.3f} ms per call back to back, {a.elapsed_time(b):.3f} ms one call by events", flush=True)
