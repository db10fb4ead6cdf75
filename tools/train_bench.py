#!/usr/bin/env python3
"""Time one training step (forward + backward through the HIP path, margin loss of the reference's demo.py:79-101,
Adam excluded) on a BASELINE workload.  Not the judged metric (bench.py measures the forward); this is the tool the
backward kernels are tuned with:  python tools/train_bench.py [--workload c3] [--steps 5]
and under  rocprofv3 --kernel-trace --stats -- python3 tools/train_bench.py  for the per-kernel split."""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from bench import WORKLOADS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--kind", default="uniform", choices=["uniform", "powerlaw"])
    ap.add_argument("--score", default="triple", choices=["triple", "edges"],
                    help="the loss's scores: score_triple(embs[src], embs[dst]) as in the reference's demo, or the fused score_edges")
    ap.add_argument("--freeze-generators", action="store_true",
                    help="diagnostics: no gradients for the weight generators' parameters (what their backward costs per step)")
    args = ap.parse_args()
    from graph_hypernetwork_forge_amd import HyperGNN, _native, synth
    _native.load()
    dev = torch.device("cuda", 0)
    cfg = WORKLOADS[args.workload]
    N, E, R, d, L, T = (cfg[k] for k in ("N", "E", "R", "d", "L", "T"))
    if args.kind == "uniform":
        ei_np, rel_np = synth.make_graph_arrays(N, E, R, cfg["seed"])
    else:
        kg = synth.make_kg(N, E, R, 1, seed=cfg["seed"], kind="powerlaw")
        ei_np, rel_np = kg.edge_index, kg.rel_ids
    names = synth.relation_names(R)
    edge_index = torch.from_numpy(ei_np).to(dev)
    rel = torch.from_numpy(rel_np).to(dev)
    x = torch.randn(N, d, generator=torch.Generator(device=dev).manual_seed(1), device=dev)
    torch.manual_seed(0)
    model = HyperGNN(text_dim=T, node_feat_dim=d, hidden_dim=d, num_layers=L).to(dev).train()
    if args.freeze_generators:
        for n_, p_ in model.named_parameters():
            if "weight_generators" in n_ or "text_encoder" in n_:
                p_.requires_grad_(False)
    src, dst = edge_index[0, :1_000_000], edge_index[1, :1_000_000]
    perm = torch.randperm(dst.numel(), device=dev)

    def step(split=None):
        model.zero_grad(set_to_none=True)
        if split is not None:
            torch.cuda.synchronize()                      # the previous step's backward is still running otherwise
        t0 = time.perf_counter()
        embs = model.forward_ids(x, edge_index, rel, names)
        if args.score == "edges":                       # the fused form (HyperGNN.score_edges); "triple": the reference's, demo.py:90-94
            pos = model.score_edges(embs, src, dst)
            neg = model.score_edges(embs, src, dst[perm])
        else:
            pos = model.score_triple(embs[src], embs[dst])
            neg = model.score_triple(embs[src], embs[dst[perm]])
        loss = torch.clamp(1.0 - pos + neg, min=0.0).mean()
        if split is not None:
            torch.cuda.synchronize()
            split.append(time.perf_counter() - t0)
        loss.backward()
        return loss

    t0 = time.time()
    loss = step()
    torch.cuda.synchronize()
    cold = time.time() - t0
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    fwd = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(fwd)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    print(json.dumps({"workload": cfg["desc"], "kind": args.kind, "score": args.score, "ms_per_train_step": ms, "forward_ms": 1e3 * sum(fwd) / len(fwd),
                      "backward_ms": ms - 1e3 * sum(fwd) / len(fwd), "edges_per_s_train": E / (ms * 1e-3),
                      "cold_step_s": cold, "loss": float(loss), "peak_hbm_gb": torch.cuda.max_memory_allocated() / 1e9}))


if __name__ == "__main__":
    main()
