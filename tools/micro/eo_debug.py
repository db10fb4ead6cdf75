"""Structured inputs for ghf_edge_outer's two-piece kernel: which (feature, column) entries come out wrong."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graph_hypernetwork_forge_amd import _native, autograd as A
from graph_hypernetwork_forge_amd.plan import build_plan
dev = torch.device("cuda:0")
d, N, E, R = 128, 64, int(os.environ.get("E", "32")), 1
rng = np.random.default_rng(0)
ei = np.stack([rng.integers(0, N, E), rng.integers(0, N, E)]).astype(np.int64)
rel = np.zeros(E, dtype=np.int64)
h = rng.standard_normal((N, d)).astype(np.float32); G = rng.standard_normal((N, d)).astype(np.float32)
t = lambda a: torch.from_numpy(a).to(dev)
plan = build_plan(t(ei), t(rel), [""], N, d, dev)
tp = A.build_train_plan(t(ei), t(rel), plan, d, dev)
dW, db = _native.edge_outer(t(h), t(G), tp.src_by_rel, tp.dst_by_rel, tp.slice_tab, tp.slice_off, R)
X = np.concatenate([h[ei[0]], h[ei[1]]], axis=1).astype(np.float64)
want = X.T @ G[ei[1]].astype(np.float64)
got = dW[0].cpu().numpy()
err = np.abs(got - want)
print("max err", err.max(), "scale", np.abs(want).max(), "db err", np.abs(db[0].cpu().numpy() - G[ei[1]].sum(0)).max())
bad = err > 1e-3 * np.abs(want).max()
print("bad fraction", bad.mean(), "bad rows (features)", np.nonzero(bad.any(1))[0][:40].tolist(), "bad cols", np.nonzero(bad.any(0))[0][:40].tolist())
# does got match want under some permutation of rows? correlate
for f in (0, 1, 5, 17, 130):
    c = [(np.abs(got[f] - want[g]).max(), g) for g in range(256)]
    print("feature row", f, "best matching want-row", min(c))
