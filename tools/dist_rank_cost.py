#!/usr/bin/env python3
"""What ONE rank of an N-GPU run costs, measured on one GPU: the rank-0 share of the sharded forward (its plan, its
chunked launches, its side-stream collectives) with every all-gather replaced by a world-1 RCCL all-gather of the
rank's own slot — the same host work and the same launches, no peer traffic.  Prints host enqueue time and device
time per forward: a lower bound of the N-GPU forward, and the place to see launch-bound behaviour before the
driver's multi-GPU run.   python tools/dist_rank_cost.py --world 8 [--chunks 4]"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from bench import WORKLOADS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--chunks", type=int, default=None)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--profile", action="store_true", help="cProfile of the host side of 20 forwards")
    args = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from graph_hypernetwork_forge_amd import HyperGNN, synth
    from graph_hypernetwork_forge_amd.dist import ShardedHyperGNN

    class OneRankOfMany(ShardedHyperGNN):
        def __init__(self, model, world, chunks):
            super().__init__(model, chunks=chunks)
            self.world, self.rank = world, 0

        def _gather_chunk(self, buf, spec, c):
            lo, hi = spec.chunk_rows(c)
            a = min(lo, max(buf.size(0) - spec.S, 0))
            mine = buf[a: a + spec.S]
            dist.all_gather_into_tensor(mine, mine)                 # world 1: same launches, no peers

    cfg = WORKLOADS[args.workload]
    N, E, R, d, L, T = (cfg[k] for k in ("N", "E", "R", "d", "L", "T"))
    ei_np, rel_np = synth.make_graph_arrays(N, E, R, cfg["seed"])
    names = synth.relation_names(R)
    edge_texts = [names[i] for i in rel_np.tolist()]
    edge_index = torch.from_numpy(ei_np).to(dev)
    x = torch.randn(N, d, device=dev)
    torch.manual_seed(0)
    model = HyperGNN(text_dim=T, node_feat_dim=d, hidden_dim=d, num_layers=L).to(dev).eval().requires_grad_(False)
    runner = OneRankOfMany(model, args.world, args.chunks)
    for _ in range(3):
        runner(x, edge_index, edge_texts)
    torch.cuda.synchronize()
    if args.profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(20):
            runner(x, edge_index, edge_texts)
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(18)
    host = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        h0 = time.perf_counter()
        runner(x, edge_index, edge_texts)
        host.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"world": args.world, "chunks": runner._spec.chunks, "rows_per_slot": runner._spec.S,
                      "ms_per_forward_one_rank": total * 1e3, "host_enqueue_ms": 1e3 * sum(host) / len(host),
                      "ideal_ms": None}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
