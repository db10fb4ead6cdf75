"""Multi-rank path on CPU: world_size 2 and 3 over gloo.

Exercises the sharding / exchange logic of graph_hypernetwork_forge_amd/dist.py (destination-range
ownership, padded shards, in-place all-gather per layer) with the oracle injected as the
per-shard compute; the result on every rank must equal the reference's golden output.
"""

import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from _util import assert_close
from graph_hypernetwork_forge_amd import HyperGNN
from graph_hypernetwork_forge_amd.dist import ShardedHyperGNN, shard_spec
from oracle import hypergnn_oracle as O


class OracleOps:
    """Same interface as dist.NativeOps, computed by the oracle on CPU tensors."""

    def __init__(self, block_nodes, split=False):
        self.bn = block_nodes
        self.split = split           # emulate a kernel that gathers a split form: here the fp32 bits + a row "scale" of 1

    def message_config(self, d, E, N=None, exact=False):
        return self.bn, 0, 48, 128

    def build_plan(self, edge_index, rel_ids, unique, N, d, device, owner=None, owner_bounds=None, edge_range=None, exact=False):
        if edge_range is not None:                                   # edge-range shards: a slice of the edge list
            keep = torch.zeros(edge_index.size(1), dtype=torch.bool)
            keep[edge_range[0]:edge_range[1]] = True
        elif owner_bounds is not None:                               # slots of unequal size, slot s -> rank s % G
            bounds, G, g = owner_bounds
            keep = (torch.bucketize(edge_index[1], bounds[1:], right=True) % G) == g
        else:
            S, G, g = owner
            keep = (edge_index[1] // S) % G == g                     # a rank owns the in-edges of its rows
        return SimpleNamespace(unique_texts=unique, ei=edge_index[:, keep], rel=rel_ids[keep], N=N, wlayout=0, E=int(keep.sum()))

    def plan_sources(self, plan):
        return plan.ei[0]

    @staticmethod
    def _params(model):
        return {k: v.detach() for k, v in model.state_dict().items()}

    def text_embs(self, model, unique, device):
        return O.text_encode(self._params(model), unique)

    def input_proj(self, model, x, out, h_split, plan):
        out.copy_(torch.relu(x @ model.input_proj.weight.t() + model.input_proj.bias))
        if h_split is not None:
            self.split_range(plan, out, h_split, 0, out.size(0))

    def all_weights(self, model, text_embs, plan, after=None):
        d = model.hidden_dim
        return [O.weight_generator(self._params(model), f"weight_generators.{l}.", text_embs, d, d)
                for l in range(model.num_layers)], None

    def split_rows(self, plan, h):
        return None

    def layer_begin(self, model, l, weights, h, plan):
        pass

    # the split exchange of dist.NativeOps with a stand-in split form: N rows of the fp32 bytes, then N float ones
    def exchanges_split(self, plan):
        return self.split

    def alloc_split(self, plan, N, d, device):
        return torch.full((N * d + N,), float("nan"))

    def split_parts(self, plan, hs, N, d):
        b = hs.view(torch.uint8)
        return [b[: N * 4 * d].view(N, 4 * d), b[N * 4 * d:].view(N, 4)]

    def split_range(self, plan, h, hs, lo, hi):
        N, d = h.shape
        hs[: N * d].view(N, d)[lo:hi] = h[lo:hi]
        hs[N * d:][lo:hi] = 1.0

    def layer_rows(self, model, l, w, h, h_split, plan, h_out, lo, hi, h_split_out=None):
        p = self._params(model)
        if self.split:                                               # other ranks' rows exist in the split form only
            N, d = h.shape
            # every row this rank's edges read must have arrived, scale included (the sparse exchange sends no others)
            read = torch.unique(plan.ei[0])
            assert bool((h_split[N * d:][read] == 1.0).all()), "a row scale did not arrive"
            full = h_split[: N * d].view(N, d)
            assert bool(torch.isfinite(full[read]).all()), "a row this rank reads did not arrive"
            assert torch.equal(full[lo:hi], h[lo:hi])
            h = full
        agg = O.message_passing_factorised(h, plan.ei, plan.rel, w["W_msg"], w["W_self"], w["bias"])
        out = O.layer_tail(agg, h, p[f"layer_norms.{l}.weight"], p[f"layer_norms.{l}.bias"])
        h_out[lo:hi] = out[lo:hi]
        if h_split_out is not None:
            self.split_range(plan, h_out, h_split_out, lo, hi)


    # edge-range shards (dist.NativeOps.layer_raw / scale_rows / tail_rows)
    def layer_raw(self, model, l, w, h, h_split, plan, partial):
        agg = O.message_passing_factorised(h, plan.ei, plan.rel, w["W_msg"], w["W_self"], w["bias"])
        cnt = torch.bincount(plan.ei[1], minlength=h.size(0)).to(agg.dtype)
        partial.copy_(agg * cnt[:, None])                            # the oracle's mean times this shard's in-degree = raw sums

    def scale_rows(self, sums, inv):
        return sums * inv[:, None]

    def tail_rows(self, model, l, agg, h, h_out, lo, hi):
        p = self._params(model)
        h_out[lo:hi] = O.layer_tail(agg[lo:hi], h[lo:hi], p[f"layer_norms.{l}.weight"], p[f"layer_norms.{l}.bias"])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case_name, block_nodes, chunks, split, ret, kw=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        (case,) = cases.graph_cases(only=[case_name])
        cfg = cases.MODELS[case.model]
        model = HyperGNN(cfg.text_dim, cfg.node_feat_dim, cfg.hidden_dim, cfg.num_layers).eval()
        model.load_state_dict({k: torch.from_numpy(v) for k, v in cfg.params().items()})
        runner = ShardedHyperGNN(model, ops=OracleOps(block_nodes, split), chunks=chunks, **(kw or {}))
        x, ei = torch.from_numpy(case.node_features), torch.from_numpy(case.edge_index)
        out = runner(x, ei, case.edge_texts)
        out2 = runner(x, ei, case.edge_texts)                         # second call reuses the shard plan
        assert torch.equal(out, out2)
        with pytest.raises(ValueError):
            runner(x, ei, case.edge_texts[:-1])
        ret[rank] = out.numpy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case_name,bn,chunks,split", [
    (2, "g3_mid32", 216, 4, False), (3, "g3_mid32", 64, 3, False), (2, "g3_mid32", 64, 1, False), (2, "g2_toy", 8, 4, False),
    (3, "g2_chain", 4, 2, False), (2, "g3_mid32", 216, 4, True), (3, "g3_mid32", 64, 3, True), (3, "g2_chain", 4, 2, True)])
def test_sharded_forward_equals_reference(golden_dir, world, case_name, bn, chunks, split):
    """split=True: the exchange moves the (stand-in) split rows and their scales, fp32 rows only after the last layer —
    the schedule dist.NativeOps runs for the default d = 128 kernel."""
    g = np.load(os.path.join(golden_dir, f"{case_name}.npz"))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), case_name, bn, chunks, split, ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == list(range(world))
    for r in range(world):
        assert_close(ret[r], g["out"], f"{case_name} world={world} rank={r}")
    for r in range(1, world):
        assert np.array_equal(ret[r], ret[0]), "all ranks must hold the same gathered result"


@pytest.mark.parametrize("world,case_name,bn,chunks,split,kw", [
    (2, "g3_mid32", 64, 3, False, dict(exchange="pairs")), (3, "g3_mid32", 216, 4, True, dict(exchange="pairs")),
    (3, "g3_mid32", 64, 3, True, dict(balance="edges")), (2, "g2_toy", 8, 4, False, dict(balance="edges")),
    (3, "g2_chain", 4, 2, True, dict(balance="edges")),
    (2, "g3_mid32", 64, 1, False, dict(mode="edges")), (3, "g3_mid32", 216, 1, False, dict(mode="edges")),
    (3, "g2_toy", 8, 1, False, dict(mode="edges")),
    # needed rows only: plan-time row lists per (chunk, peer), gather-pack -> send/recv -> scatter; the last layer whole
    (2, "g3_mid32", 64, 3, False, dict(exchange="sparse")), (3, "g3_mid32", 216, 4, True, dict(exchange="sparse")),
    (3, "g2_chain", 4, 2, True, dict(exchange="sparse")), (3, "g3_mid32", 64, 3, True, dict(exchange="sparse", balance="edges")),
    (2, "g2_toy", 8, 4, False, dict(exchange="sparse"))])
def test_sharded_variants_equal_reference(golden_dir, world, case_name, bn, chunks, split, kw):
    """The pairwise exchange (every rank sends its slot straight to each peer), slots balanced by in-edge count (unequal
    sizes, pairwise exchange), and the north-star split — edge-range shards, raw partial sums reduced across ranks, tail on
    the owned rows, all-gather — all reproduce the reference on every rank (g3_mid32 has a 510-in-degree hub)."""
    g = np.load(os.path.join(golden_dir, f"{case_name}.npz"))
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), case_name, bn, chunks, split, ret, kw), nprocs=world, join=True)
    assert sorted(ret.keys()) == list(range(world))
    for r in range(world):
        assert_close(ret[r], g["out"], f"{case_name} world={world} rank={r} {kw}")
    for r in range(1, world):
        assert np.array_equal(ret[r], ret[0]), "all ranks must hold the same gathered result"


def test_shard_spec_balanced_by_edges():
    """Power-law in-degrees: equal-row slots leave one rank with the hub's edges; equal-cost cuts do not."""
    from graph_hypernetwork_forge_amd import synth
    N, E, bn, world, chunks = 200_000, 2_000_000, 216, 8, 4
    ei, _ = synth.make_graph_arrays(N, E, 16, seed=1005, kind="powerlaw")
    nb = -(-N // bn)
    be = np.bincount(ei[1] // bn, minlength=nb)
    per_rank = lambda specs: [sum(int(be[lo // bn: -(-hi // bn)].sum()) for lo, hi in s.owned()) for s in specs]   # noqa: E731
    rows = per_rank([shard_spec(N, bn, world, r, chunks) for r in range(world)])
    specs = [shard_spec(N, bn, world, r, chunks, be) for r in range(world)]
    bal = per_rank(specs)
    assert sum(rows) == sum(bal) == E
    # no cut can go below the heaviest block (the hub's); beyond that the ranks are level, which equal-row slots are not
    assert max(bal) <= max(1.15 * E / world, be.max() * 1.05) and max(bal) < max(rows), (rows, bal, int(be.max()))
    assert sorted(bal)[-2] < 1.15 * E / world
    covered = np.zeros(N, dtype=np.int32)
    for s in specs:
        assert not s.uniform and s.bounds[0] == 0 and s.bounds[-1] >= N and all(b % bn == 0 for b in s.bounds)
        for lo, hi in s.owned():
            covered[lo:hi] += 1
    assert (covered == 1).all()


def test_shard_spec_covers_all_rows_once():
    for N, bn, world, chunks in [(1_000_000, 216, 8, 4), (1000, 216, 8, 4), (5, 4, 3, 2), (4_000_000, 88, 8, 1),
                                 (217, 216, 2, 4), (100_000, 216, 4, 16)]:
        specs = [shard_spec(N, bn, world, r, chunks) for r in range(world)]
        s0 = specs[0]
        assert s0.S % bn == 0 and s0.padded_rows >= N and 1 <= s0.chunks <= chunks
        covered = np.zeros(N, dtype=np.int32)
        for s in specs:
            for lo, hi in s.owned():
                assert lo % bn == 0
                covered[lo:hi] += 1
                owner = (np.arange(lo, hi) // s.S) % world
                assert (owner == s.rank).all()                         # the plan's ownership rule
        assert (covered == 1).all()
        for c in range(s0.chunks):                                      # a chunk's slots tile its all-gather slice
            lo, hi = s0.chunk_rows(c)
            assert hi - lo == world * s0.S


def _plan_worker(rank, world, port, N, E, R, bn, chunks, ret):
    """One rank of the eight: everything ShardedHyperGNN.plan_for does at BASELINE config 3's size — shard geometry, this rank's
    edges, the sparse exchange's row lists through the group — and nothing of the forward."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graph_hypernetwork_forge_amd import synth
        ei_np, rel = synth.make_graph_arrays(N, E, R, seed=1003)
        names = synth.relation_names(R)
        texts = [names[r] for r in rel.tolist()]
        model = SimpleNamespace(hidden_dim=128)
        for kw in (dict(exchange="sparse"), dict(exchange="sparse", balance="edges"), dict(mode="edges")):
            runner = ShardedHyperGNN(model, ops=OracleOps(bn), chunks=chunks, **kw)
            plan = runner.plan_for(torch.from_numpy(ei_np), texts, N, torch.device("cpu"))
            spec, sp = runner._spec, runner._sparse
            info = dict(E=plan.E, owned=sum(hi - lo for lo, hi in spec.owned()), chunks=spec.chunks)
            if sp is not None:
                info.update(needed=sp["rows_needed"], other=sp["rows_other"],
                            sent=sum(int(t.numel()) for c in sp["send"] for t in c), recv=sum(int(t.numel()) for c in sp["recv"] for t in c))
                for c in range(spec.chunks):
                    for p in range(world):
                        lo, hi = spec.slot(c, p)
                        t = sp["recv"][c][p]
                        assert p != rank or t.numel() == 0
                        assert t.numel() == 0 or (int(t.min()) >= lo and int(t.max()) < hi), "a row asked of the wrong owner"
            ret[(rank, tuple(sorted(kw.items())))] = info
    finally:
        dist.destroy_process_group()


def test_eight_ranks_plan_config_3_without_computing():
    """The first 8-GPU run must not die in planning: eight gloo ranks build BASELINE config 3's shards (1 M nodes, 10 M edges,
    blocks of 384 nodes, four chunks) — row-balanced and edge-balanced destination shards with the sparse exchange's row lists,
    and the north-star edge ranges — and check what they agreed on.  Plan only: no layer runs."""
    N, E, R, bn, chunks, world = 1_000_000, 10_000_000, 64, 384, 4, 8
    ret = mp.Manager().dict()
    mp.spawn(_plan_worker, args=(world, _free_port(), N, E, R, bn, chunks, ret), nprocs=world, join=True)
    for kw in (dict(exchange="sparse"), dict(exchange="sparse", balance="edges"), dict(mode="edges")):
        key = tuple(sorted(kw.items()))
        infos = [ret[(r, key)] for r in range(world)]
        assert sum(i["E"] for i in infos) == E                        # every edge on exactly one rank
        if "mode" in kw:
            assert max(i["E"] for i in infos) - min(i["E"] for i in infos) <= 1
            continue
        assert sum(i["owned"] for i in infos) == N                    # every row owned once
        assert sum(i["sent"] for i in infos) == sum(i["recv"] for i in infos) == sum(i["needed"] for i in infos)
        for i in infos:
            # a uniform random shard of 1.25 M edges reads ~71 % of the other ranks' rows (1 - exp(-1.25)); never more than all
            assert 0.6 * i["other"] < i["needed"] <= i["other"], i
            # (equal-row slots of whole blocks: 82 blocks per slot cover 1,007,616 rows, so the last rank's last slot is short — it
            # holds 5 % fewer edges than the others at this size)
            assert abs(i["E"] - E / world) < 0.08 * E / world, i
