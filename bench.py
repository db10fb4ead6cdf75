#!/usr/bin/env python3
"""bench.py — edges/s of HyperGNN.forward on MI355X (BASELINE.json metric), one JSON line on rank 0.

Workload (config.workload): BASELINE config 3 — synthetic uniform KG, 1 M nodes / 10 M edges /
64 relation types, hidden_dim = 128, 3 layers, text_dim = 64, node_feat_dim = 128, fp32.
A "step" is one full forward (text encoding, input projection, 3 x (weight generation +
message layer with fused tail)) with the graph plan cached and inputs resident in HBM.
N > 1 (launched by torch.distributed.run): the SAME graph sharded by destination range,
one rank per GPU, all-gather of h per layer over RCCL -> "scaling": "strong".

Extra objects on the line:
  roofline     — the dominant kernel (message layer): achieved = algorithmic flops / mean
                 launch duration (HIP events on the launch stream), against the fp32 matrix
                 peak, which binds this kernel before HBM does; the HBM
                 view (algorithmic bytes / duration vs 8 TB/s) is reported beside it.
  cpu_baseline — the oracle (CPU restatement of the reference op sequence) timed on the host
                 cores on a bounded sample of the same workload.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

WORKLOADS = {
    # name: N, E, R, d, L, T (F: node feature width when it differs from d)
    # BASELINE config 1: the reference's own demo (demo.py:49-66) — the ToyKnowledgeGraph fixture, launch-latency bound
    "c1": dict(N=8, E=11, R=7, d=32, L=2, T=64, F=16, seed=42, kind="toy",
               desc="ToyKnowledgeGraph 8 nodes / 11 edges / 7 rel, feat 16, hidden 32, L=2 (BASELINE config 1, demo.py:49-66)"),
    "c3": dict(N=1_000_000, E=10_000_000, R=64, d=128, L=3, T=64, seed=1003,
               desc="synthetic uniform KG 1M nodes / 10M edges / 64 rel, hidden 128, L=3 (BASELINE config 3)"),
    "c2": dict(N=100_000, E=1_000_000, R=32, d=64, L=2, T=64, seed=1002,
               desc="synthetic uniform KG 100k nodes / 1M edges / 32 rel, hidden 64, L=2 (BASELINE config 2)"),
    # not a default: 64 GB of per-edge results on one GPU, a 64M-entry edge_texts list on the host
    "c5": dict(N=4_000_000, E=64_000_000, R=256, d=256, L=4, T=64, seed=1005, kind="powerlaw",
               desc="synthetic power-law KG 4M nodes / 64M edges / 256 rel, hidden 256, L=4 (BASELINE config 5)"),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_MATRIX_PEAK_TF = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_* peak (= fp32 vector peak)
F16_MATRIX_PEAK_TF = 2500.0    # MI355X_MICROARCH.md: dense bf16/fp16 MFMA peak
L2_STREAM_CEILING_GBS = 33000.0  # tools/micro/l2stream.hip on this chip: L2-resident streaming into registers, all 256 CUs


def layer_bytes(N, E, R, d):
    """SURVEY.md §8d algorithmic bytes of one message layer + tail."""
    return E * (4 * d + 24) + N * 8 * d + R * (2 * d * d + d) * 4


def layer_flops(N, E, R, d):
    """SURVEY.md §8d algorithmic flops of one message layer + tail."""
    return E * (4 * d * d + 2 * d) + 10 * N * d


def kern_name(plan, d):
    from graph_hypernetwork_forge_amd import _native
    return {_native.WLAYOUT_SPLIT2H: "message_bx_kernel"}.get(
        plan.wlayout, "message_pp_kernel" if plan.block_nodes > 1 else "message_generic_kernel")


PMC_FILES = {"c3": "r04_message_kernel_pmc.json", "c2": "r04_c2_kernel_pmc.json", "c5": "r04_c5_kernel_pmc.json"}
PMC_SOURCES = {"c3": ["message_bx.hip"], "c2": ["message_bx.hip", "message_pp.hip"], "c5": ["message_rs.hip"]}


def source_digest(files) -> str:
    """sha256 over the kernel sources a PMC file was measured on (tools/pmc.sh stores it; a profile of other code is stale)."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "graph-hypernetwork-forge_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, budget_s=12.0, full=None):
    """Time the oracle on the host cores: (i) the reference op sequence (per-edge [E,d,d] weight copies) on a bounded,
    scaled-down sample — the reference cannot hold the full workload (2.6 TB at C3); (ii) the factorised restatement
    (per-relation loop, O(E d) memory; oracle/hypergnn_oracle.py, validated against the reference through the golden
    vectors) on the FULL workload, `full` = (node_features, edge_index, edge_texts) as numpy / list."""
    from graph_hypernetwork_forge_amd import synth
    from oracle import hypergnn_oracle as O
    d, R, L, T = cfg["d"], cfg["R"], cfg["L"], cfg["T"]
    if cfg.get("kind") == "toy":                           # config 1: the reference's demo forward on the same toy graph
        xf, eif, tf = full
        params = synth.hypergnn_params(T, cfg["F"], d, L, seed=7)
        # an 11-edge graph is all fixed cost: with every core of a 128-thread host the intra-op thread pool makes it slower
        # (37 ms per forward against ~1 ms on one thread) — both are timed, the faster one is the baseline
        best, all_threads = None, torch.get_num_threads()
        for threads in sorted({1, all_threads}):
            torch.set_num_threads(threads)
            for _ in range(10):
                O.forward(params, xf, eif, tf, variant="reference")
            t0, n = time.time(), 0
            while time.time() - t0 < min(budget_s, 3.0):
                O.forward(params, xf, eif, tf, variant="reference")
                n += 1
            t = (time.time() - t0) / n
            if best is None or t < best[0]:
                best = (t, threads, n)
        torch.set_num_threads(all_threads)
        t, threads, n = best
        return {"value": len(tf) / t, "unit": "edges/s", "cores": threads, "cpu_model": cpu_model(), "kind": "port",
                "sample": f"oracle forward (reference op sequence) on the ToyKnowledgeGraph itself, mean of {n} runs on {threads} "
                          f"thread(s) (the faster of 1 and {all_threads} threads)",
                "ms_per_forward": t * 1e3}
    # the reference materialises 4*E*d^2*4 bytes: size the sample to ~8 GB of that
    E = int(min(cfg["E"], max(2000, 8e9 / (16 * d * d))))
    N = max(100, E // 10)
    kg = synth.make_kg(N, E, R, d, seed=cfg["seed"])
    params = synth.hypergnn_params(T, d, d, L, seed=7)
    texts = kg.edge_texts()
    threads = torch.get_num_threads()
    times = []
    t_start = time.time()
    for i in range(4):
        t0 = time.time()
        O.forward(params, kg.node_features, kg.edge_index, texts, variant="reference")
        dt = time.time() - t0
        if i > 0:
            times.append(dt)
        if time.time() - t_start > budget_s and times:
            break
    t = float(np.median(times))
    t0 = time.time()
    O.forward(params, kg.node_features, kg.edge_index, texts, variant="factorised")
    t_fact = time.time() - t0
    res = {"value": E / t, "unit": "edges/s", "cores": threads, "cpu_model": cpu_model(), "kind": "port",
           "sample": f"oracle forward (reference op sequence incl. per-edge [E,d,d] weight copies) on a scaled-down "
                     f"instance of the workload: N={N}, E={E}, R={R}, d={d}, L={L}; median of {len(times)} runs",
           "ms_per_forward": t * 1e3, "factorised_variant_edges_per_s": E / t_fact}
    if full is not None:
        xf, eif, tf = full
        paramsf = synth.hypergnn_params(T, xf.shape[1], d, L, seed=7)
        runs = []
        for _ in range(2):                                 # one warm-up (page faults, thread pool), one timed; a slow host
            t0 = time.time()                               # (> 40 s per forward) keeps its only run
            O.forward(paramsf, xf, eif, tf, variant="factorised")
            runs.append(time.time() - t0)
            if runs[-1] > 40.0:
                break
        res["full_size_factorised_edges_per_s"] = len(tf) / runs[-1]
        res["full_size_factorised_s_per_forward"] = runs[-1]
        res["full_size_sample"] = (f"oracle forward, factorised variant, on the whole workload (N={xf.shape[0]}, E={len(tf)}, "
                                   f"R={R}, d={d}, L={L}), string -> id mapping included as in the reference; "
                                   f"{'second of two runs' if len(runs) == 2 else 'single run'}")
    return res


def roofline_note(wide: bool, plan, d: int) -> str:
    """What the roofline figures of this workload's dominant kernel mean (which bound is the real one)."""
    from graph_hypernetwork_forge_amd import _native
    head = "judged against HBM as BASELINE asks; "
    if wide:
        return head + ("the relation-stationary layer: pass 0 (runs' source rows summed) and pass 2 (segment sums + tail) are gather / stream "
                       "bound, pass 1 gathers both rows of every run AND does three fp16 products per flop on one wave per SIMD (293 "
                       "registers) — at BASELINE config 5 it runs at ~75 % of the dense fp16 matrix peak (DESIGN.md §3, message_rs)")
    if plan.wlayout == _native.WLAYOUT_SPLIT2H and d == 128:
        return head + ("the block kernel is bound inside the CU: the vector-memory return path (TD) is busy ~80 % of the launch — a "
                       "relation's weights re-streamed from L2 per (block, relation) chunk plus the gathered rows, half of it waiting for "
                       "L2 / Infinity-Cache data — and the matrix pipe ~34 % (profiles/r04_message_kernel_pipes.json, DESIGN.md §3)")
    if plan.wlayout == _native.WLAYOUT_SPLIT2H:
        return head + ("hidden 64: a quarter of the matrix work per row, so the fixed cost per (block, relation) chunk — barrier, "
                       "hand-shakes, descriptor pipeline, two DMA round trips: ~7 k cycles — decides; two workgroups per CU overlap "
                       "their chains (DESIGN.md §3, Hidden 64)")
    return head + "a graph this small is one launch latency per kernel: see the HIP-graph replay line (bench.py --hip-graph)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hip-graph", action="store_true",
                    help="single GPU: a step replays the captured HIP graph of the warm forward (HyperGNN.graphed)")
    ap.add_argument("--kernel-reps", type=int, default=10, help="timed launches of the message kernel for the roofline")
    ap.add_argument("--dist-mode", default=os.environ.get("GHF_DIST_MODE", "dst"), choices=["dst", "edges"],
                    help="N > 1: destination shards + all-gather (default) or edge-range shards + reduction (BASELINE config 4 as written)")
    ap.add_argument("--exchange", default=os.environ.get("GHF_DIST_EXCHANGE", "auto"), choices=["auto", "allgather", "pairs", "sparse"],
                    help="N > 1, dst mode: all_gather_into_tensor per chunk, pairwise send/recv, needed rows only (plan-time row "
                         "lists, gather-pack / send / scatter), or the fastest of the three (timed during the warm-up)")
    ap.add_argument("--balance", default=os.environ.get("GHF_DIST_BALANCE", "rows"), choices=["rows", "edges"])
    ap.add_argument("--weak", action="store_true", help="N > 1: one full workload per GPU (nodes and edges scale with N)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("GHF_DIST_BACKEND", "nccl")          # "gloo": rehearsal of the multi-rank path on one card
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    from graph_hypernetwork_forge_amd import HyperGNN, _native, synth
    from graph_hypernetwork_forge_amd.dist import ShardedHyperGNN

    _native.load()                                       # fail loudly if the HIP library is missing
    cfg = dict(WORKLOADS[args.workload])
    if args.weak and world > 1:
        cfg["N"], cfg["E"] = cfg["N"] * world, cfg["E"] * world
        cfg["desc"] += f" x{world} (weak scaling: one such graph per GPU)"
    N, E, R, d, L, T = (cfg[k] for k in ("N", "E", "R", "d", "L", "T"))

    t0 = time.time()
    F = cfg.get("F", d)
    if cfg.get("kind") == "toy":
        from graph_hypernetwork_forge_amd import ToyKnowledgeGraph
        kg = ToyKnowledgeGraph(feat_dim=F)
        ei_np, edge_texts = kg.edge_index.numpy(), list(kg.edge_texts)
        edge_index, x = kg.edge_index.to(dev), kg.node_features.to(dev)
    else:
        ei_np, rel_np = synth.make_graph_arrays(N, E, R, cfg["seed"], cfg.get("kind", "uniform"))
        names = synth.relation_names(R)
        edge_texts = [names[i] for i in rel_np.tolist()]
        edge_index = torch.from_numpy(ei_np).to(dev)
        gen = torch.Generator(device=dev).manual_seed(cfg["seed"])
        x = torch.randn(N, F, generator=gen, device=dev)     # throughput is value-independent
    torch.manual_seed(0)
    model = HyperGNN(text_dim=T, node_feat_dim=F, hidden_dim=d, num_layers=L).to(dev).eval().requires_grad_(False)
    t_setup = time.time() - t0

    runner = None
    exchange_pick = None
    exchange_why = None
    if world > 1:
        os.environ.setdefault("GHF_DIST_ROW_STATS", "1")      # plan-time row lists also for the full exchanges: needed vs received rows
        exch = args.exchange if args.dist_mode == "dst" and (args.balance == "rows" or args.exchange == "sparse") else (
            "pairs" if args.balance == "edges" else "allgather")
        if exch == "auto":
            # both exchanges are built and timed for a few forwards; every rank takes the same decision (max over ranks)
            cand = {}
            for name in ("allgather", "pairs", "sparse"):
                r = ShardedHyperGNN(model, mode="dst", exchange=name, balance="rows")
                with torch.no_grad():
                    r(x, edge_index, edge_texts)
                    dist.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(3):
                        r(x, edge_index, edge_texts)
                    torch.cuda.synchronize(); dt = time.perf_counter() - t0
                tt = torch.tensor([dt], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                cand[name] = (float(tt.item()) / 3 * 1e3, r)
            exch = min(cand, key=lambda k: cand[k][0])
            exchange_pick = {k: v[0] for k, v in cand.items()}
            runner = cand[exch][1]
            exchange_why = (f"auto: the fastest of three timed forwards each (max over ranks): "
                            + ", ".join(f"{k} {v:.2f} ms" for k, v in exchange_pick.items()) + f" -> {exch}")
        else:
            runner = ShardedHyperGNN(model, mode=args.dist_mode, exchange=exch, balance=args.balance)
            exchange_why = (f"--exchange {args.exchange}" if args.exchange != "auto" else
                            "edge-balanced slots differ in size: pairwise messages" if args.balance == "edges" else
                            "edge-range mode: reduce-scatter + all-gather") + f" -> {runner.exchange}"

    graphed = model.graphed(x, edge_index, edge_texts) if (args.hip_graph and world == 1) else None

    def step():
        if graphed is not None:
            return graphed.replay()
        with torch.no_grad():
            return runner(x, edge_index, edge_texts) if runner else model(x, edge_index, edge_texts)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # cold forward: string -> id mapping + plan build + first launch
    sync(); t0 = time.time(); out = step(); sync()
    t_cold = time.time() - t0
    assert out.shape == (N, d) and (bool(torch.isfinite(out[:1024]).all()) or os.environ.get("GHF_VARIANT", "").startswith(("ablate", "exp")))

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_step = elapsed / args.steps * 1e3

    dist_detail = None
    if world > 1:
        # where a forward goes on every rank: the same forward with the exchange switched off (compute only), and with the
        # kernels switched off (exchange only); "exposed" = what the full forward takes beyond its compute
        def timed(profile, reps=3):
            runner.profile = profile
            step(); sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps * 1e3
            runner.profile = "full"
            return dt
        comp, exch_ms = timed("compute"), timed("exchange")
        step(); sync()
        nbytes = runner.stats.get("bytes_recv", 0.0)
        sp = runner._sparse or {}
        rows_needed = float(sp.get("rows_needed", -1))             # rows of other ranks this rank's edges read, per layer
        rows_recv = rows_needed if runner.exchange == "sparse" else float(sp.get("rows_other", -1))   # (the last layer's rows travel whole)
        mine = torch.tensor([comp, exch_ms, max(0.0, ms_step - comp), nbytes, rows_needed, rows_recv], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        links = max(1, world - 1)                         # xGMI: one link per peer
        dist_detail = {
            "mode": runner.mode, "exchange": runner.exchange, "balance": runner.balance, "chunks": runner._spec.chunks,
            "exchange_candidates_ms": exchange_pick, "exchange_chosen_because": exchange_why,
            "per_rank": [{"rank": r, "compute_ms": float(v[0]), "exchange_ms": float(v[1]), "exposed_exchange_ms": float(v[2]),
                          "bytes_received_per_forward": float(v[3]),
                          "rows_needed_per_layer": int(v[4]), "rows_received_per_layer": int(v[5]),
                          "gb_per_s_per_link": float(v[3]) / links / max(float(v[1]), 1e-9) / 1e6} for r, v in enumerate(allr)],
            "note": "compute_ms / exchange_ms: the forward with the exchange / the kernels switched off (3 runs each); "
                    "exposed = ms_per_step - compute_ms; per-link rate = bytes received / (world - 1) links / exchange_ms; "
                    "rows_needed = distinct source rows of this rank's in-edges that other ranks own (the exchange's floor per "
                    "layer), rows_received = what this exchange delivers per inner layer (-1: edge-range mode)"}

    line = None
    if rank == 0:
        # ---- dominant kernel: one message layer (K2+K3), timed alone with HIP events on its stream ----
        plan = model.plan_for(edge_index, edge_texts, N, dev) if world == 1 else runner.plan_for(edge_index, edge_texts, N, dev)
        with torch.no_grad():
            te = model.text_encoder(plan.unique_texts, dev)
            W, W_self, bias = model.weight_generators[0].generate(te, plan.wlayout)
            h = _native.input_proj_fwd(x, model.input_proj.weight, model.input_proj.bias)
        h_out = torch.empty_like(h)
        # kernels that gather pre-split rows get them as in the forward (made by the previous layer's tail)
        hs = _native.split_rows(h, plan.wlayout) if plan.wlayout in _native.SPLIT_LAYOUTS else None
        ln = model.layer_norms[0]
        edge_sharded = world > 1 and runner.mode == "edges"           # this rank's edges reach every row
        slots = [(0, N)] if world == 1 or edge_sharded else runner._spec.owned()     # this rank's destination rows
        wide = plan.block_nodes == 1 and _native.rs_supported(d)     # relation-stationary layer (csrc/message_rs.hip)
        if wide:
            from graph_hypernetwork_forge_amd.plan import build_rs
            plan.rs = plan.rs or build_rs(plan)
            Y = plan.rs.scratch(plan.E, d, dev)

        def msg():
            if wide:
                _native.edge_transform_fwd(h, plan.rs, W, W_self, bias, Y)        # incl. cutting h and the weights into pieces
                for lo, hi in slots:
                    _native.segment_tail_fwd(Y, plan.rs, h, ln.weight, ln.bias, ln.eps, h_out, row0=lo, rows=hi - lo)
                return
            for lo, hi in slots:
                _native.message_layer_fwd(h, plan, W, W_self, bias, plan.wlayout, ln.weight, ln.bias, ln.eps, h_out,
                                          row0=lo, rows=hi - lo, h_split=hs)
        for _ in range(2):
            msg()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.kernel_reps)]
        for a, b in evs:
            a.record(); msg(); b.record()
        torch.cuda.synchronize()
        k_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        share = plan.E / E if edge_sharded else sum(hi - lo for lo, hi in slots) / N   # this rank's share of the layer (1.0 at N=1)
        flops = layer_flops(N, E, R, d) * share
        byts = layer_bytes(N, E, R, d) * share
        tf = flops / (k_ms * 1e-3) / 1e12
        gbs = byts / (k_ms * 1e-3) / 1e9
        # HBM-side traffic per launch from the committed rocprofv3 PMC passes of this same command (separate
        # --pmc runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as read)
        traffic, traffic_src = None, None
        pmc_name = PMC_FILES.get(args.workload, "")
        pmc_path = os.path.join(ROOT, "profiles", pmc_name)
        pmc = {}
        if world == 1 and pmc_name and os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            # counters of an earlier build of the kernels say nothing about this one: tools/pmc.sh stores a
            # digest of the kernel sources it measured; the traffic figure is dropped when the sources have moved on
            if pmc.get("_source_sha256") != source_digest(PMC_SOURCES[args.workload]):
                traffic_src = (f"none: profiles/{pmc_name} was measured on other kernel sources (digest "
                               f"{pmc.get('_source_sha256')}); rerun tools/pmc.sh")
                pmc = {}
            names = [pmc.get("_kernel", "")] if "kernels" not in pmc else list(pmc["kernels"])
            want = ["edge_transform", "segment_tail"] if (plan.block_nodes == 1 and _native.rs_supported(d)) else [kern_name(plan, d)]
            if not all(any(w_ in n for n in names) for w_ in want):
                pmc = {}                                    # counters of another kernel: not this run's traffic
        if pmc:
            layer_kernels = ("edge_transform", "segment_tail", "segment_partial", "run_rows", "split2h_rows", "rs_w")   # one launch each per layer
            parts = [pmc] if "kernels" not in pmc else [v for k, v in pmc["kernels"].items() if any(n in k for n in layer_kernels)]
            traffic = sum((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 for v in parts)
            traffic_src = f"profiles/{pmc_name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, KiB; FETCH x2; per-launch means" + \
                          (", summed over the layer's kernels)" if len(parts) > 1 else ")")
        kern = {_native.WLAYOUT_SPLIT2H: kern_name(plan, d) + "<%d>" % d}.get(
            plan.wlayout, "message_pp_kernel<%d>" % d if plan.block_nodes > 1 else "message_generic_kernel")
        if wide:
            kern = ("run_rows_kernel + " if plan.rs is not None and plan.rs.run_start is not None else "") + \
                   "edge_transform%s_kernel + segment_tail_kernel<%d> (one layer, cutting rows and weights included)" % ("" if _native.rs_exact() else "_h", d // 64)
        # matrix work the kernel issues per algorithmic flop: 3 fp16 products (bx), 1 fp32 (pp)
        prod, mpeak = {_native.WLAYOUT_SPLIT2H: (3, F16_MATRIX_PEAK_TF)}.get(
            plan.wlayout, (1, FP32_MATRIX_PEAK_TF))
        if wide and not _native.rs_exact():
            prod, mpeak = 3, F16_MATRIX_PEAK_TF
        l2_bytes = None
        if traffic is not None and "TCC_REQ_sum" in pmc:      # (single-kernel layers)
            l2_bytes = pmc["TCC_REQ_sum"] * 128.0
        roofline = {"kernel": kern,
                    "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "ms_per_launch": k_ms, "algorithmic_flops_per_launch": flops,
                    "algorithmic_bytes_per_launch": byts,
                    "mfma": {"achieved": tf, "unit": "TFLOP/s (algorithmic fp32 flops)", "products_per_flop": prod,
                             "peak": mpeak / prod, "frac": tf / (mpeak / prod)},
                    "l2_to_cu": None if l2_bytes is None else {
                        "bytes_per_launch": l2_bytes, "achieved": l2_bytes / (k_ms * 1e-3) / 1e9,
                        "ceiling": L2_STREAM_CEILING_GBS, "unit": "GB/s",
                        "source": "TCC_REQ_sum x 128 B (same PMC file); ceiling measured by tools/micro/l2stream.hip"},
                    "note": roofline_note(wide, plan, d)}
        line = {
            "metric": "edges/s (HyperGNN forward)", "value": E / (ms_step * 1e-3), "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": cfg["desc"], "nodes": N, "edges": E, "relations": R, "hidden_dim": d,
                       "layers": L, "text_dim": T, "plan": "cached (warm)",
                       "launch": "captured HIP graph replay" if graphed is not None else "one C-ABI call per kernel",
                       "parallelism": "single GPU" if world == 1 else (
                           f"block-cyclic destination shards x{world} ({runner.balance}-balanced), chunked {runner.exchange} exchange of h per layer overlapped with compute"
                           if runner.mode == "dst" else f"edge-range shards x{world}, reduce-scatter of partial sums + all-gather per layer")},
            "cold_forward_ms": t_cold * 1e3, "setup_s": t_setup,
            "whole_forward_hbm_gbs": (L * layer_bytes(N, E, R, d) + 2 * N * d * 4) / (ms_step * 1e-3) / 1e9,
            "roofline": roofline,
        }
        if dist_detail is not None:
            line["dist"] = dist_detail
            line["scaling"] = "weak" if args.weak else "strong"
        if world == 1 and not args.no_cpu_baseline:
            full = (x.cpu().numpy(), ei_np, edge_texts) if args.workload != "c5" else None    # (C5: 64M strings, 16 GB of h)
            line["cpu_baseline"] = cpu_baseline(cfg, full=full)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
