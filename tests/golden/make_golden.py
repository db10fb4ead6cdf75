#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference on the inputs of cases.py.

Run in the build container only (the reference lives at /root/reference and never
travels):   cd /tmp && python /root/repo/tests/golden/make_golden.py [case ...]

What is stored is data: inputs that cannot be regenerated (the toy-KG features the
reference draws with torch.randn), the reference's outputs, a few intermediates,
and a sha256 of the synthetic parameters.  No reference source or bytecode is stored.
"""

from __future__ import annotations

import os
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import cases  # noqa: E402
from graph_hypernetwork_forge import HyperGNN, ToyKnowledgeGraph, WeightGenerator  # noqa: E402  (the reference)


def load_params(module: torch.nn.Module, params) -> None:
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()}
    missing, unexpected = module.load_state_dict(sd, strict=True)
    assert not missing and not unexpected


def ref_model(cfg: cases.ModelCfg) -> HyperGNN:
    m = HyperGNN(text_dim=cfg.text_dim, node_feat_dim=cfg.node_feat_dim, hidden_dim=cfg.hidden_dim,
                 num_layers=cfg.num_layers, dropout=0.0, char_emb_dim=cfg.char_emb_dim)
    load_params(m, cfg.params())
    return m.eval()


def run_graph_case(case: cases.GraphCase) -> dict:
    cfg = cases.MODELS[case.model]
    model = ref_model(cfg)
    x = torch.from_numpy(case.node_features)
    ei = torch.from_numpy(case.edge_index)
    rec = {}
    hooks = []
    if case.intermediates:
        inter = {}

        def grab(name):
            def fn(_m, _inp, out):
                inter.setdefault(name, []).append(out)
            return fn
        hooks.append(model.text_encoder.register_forward_hook(grab("text_embs")))
        for l, (gen, ln) in enumerate(zip(model.weight_generators, model.layer_norms)):
            hooks.append(gen.register_forward_hook(grab(f"wg{l}")))
            hooks.append(ln.register_forward_hook(grab(f"h{l + 1}")))
    t0 = time.time()
    with torch.no_grad():
        out = model(x, ei, case.edge_texts)
    dt = time.time() - t0
    for hk in hooks:
        hk.remove()
    out_np = out.numpy()
    rec["params_sha256"] = np.array(cases.params_digest(cfg.params()))
    rec["out_l2"] = np.array(float(np.linalg.norm(out_np.astype(np.float64))))
    rec["out_sum"] = np.array(float(out_np.astype(np.float64).sum()))
    rec["out_shape"] = np.array(out_np.shape)
    if case.store == "sampled":
        rows = np.arange(0, out_np.shape[0], cases.SAMPLE_STRIDE)
        rec["rows"] = rows
        rec["out_rows"] = out_np[rows]
    else:
        rec["out"] = out_np
    if case.intermediates:
        rec["text_embs"] = inter["text_embs"][0].numpy()
        for l in range(cfg.num_layers):
            w = inter[f"wg{l}"][0]
            rec[f"W_msg{l}"], rec[f"W_self{l}"], rec[f"bias{l}"] = (w[k].numpy() for k in ("W_msg", "W_self", "bias"))
            rec[f"h{l + 1}"] = inter[f"h{l + 1}"][0].numpy()
    print(f"  {case.name}: N={x.shape[0]} E={ei.shape[1]} out={tuple(out.shape)} ref forward {dt:.2f}s", flush=True)
    return rec


def run_wg_case(c: cases.WGCase) -> dict:
    gen = WeightGenerator(text_dim=c.text_dim, d_in=c.d_in, d_out=c.d_out, hidden_dim=c.hidden_dim,
                          num_hidden=c.num_hidden, dropout=c.dropout)
    load_params(gen, c.params())
    gen.eval()
    with torch.no_grad():
        out = gen(torch.from_numpy(c.text_emb()))
    rec = {k: (v.numpy() if c.keep is None else v.numpy()[list(c.keep)]) for k, v in out.items()}
    rec["params_sha256"] = np.array(cases.params_digest(c.params()))
    print(f"  {c.name}: " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in out.items()), flush=True)
    return rec


def main(argv) -> None:
    torch.set_num_threads(8)
    only = argv[1:] or None
    kg = ToyKnowledgeGraph(feat_dim=16)           # the reference's own fixture (BASELINE config 1 input)
    toy_path = os.path.join(HERE, "toy_features.npz")
    if only is None or not os.path.exists(toy_path):
        np.savez(toy_path, node_features=kg.node_features.numpy(), edge_index=kg.edge_index.numpy(),
                 edge_texts=np.array(kg.edge_texts))
        assert kg.edge_index.numpy().tolist() == cases.toy_edge_index().tolist()
        assert kg.edge_texts == cases.toy_edge_texts()
    for name in cases.GRAPH_CASE_NAMES:
        if only and name not in only:
            continue
        (case,) = cases.graph_cases(kg.node_features.numpy(), only=[name])
        np.savez(os.path.join(HERE, f"{name}.npz"), **run_graph_case(case))
    if only is None or "wg" in only:
        recs = {}
        for c in cases.WG_CASES:
            for k, v in run_wg_case(c).items():
                recs[f"{c.name}/{k}"] = v
        np.savez(os.path.join(HERE, "wg_cases.npz"), **recs)


if __name__ == "__main__":
    main(sys.argv)
