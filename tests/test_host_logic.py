"""Host-side logic that needs no GPU: C-ABI exports, the module mirror, plan cache keys, synthetic data."""

import ctypes
import os
import re

import numpy as np
import pytest
import torch

import cases
from graph_hypernetwork_forge_amd import HyperGNN, TextEncoder, ToyKnowledgeGraph, WeightGenerator, _build, _native, synth
from graph_hypernetwork_forge_amd.plan import PlanCache, relation_ids
from oracle import hypergnn_oracle as O


# ---- C ABI ------------------------------------------------------------------------------------

def test_library_exports_every_symbol_of_the_header():
    """The .so loads on a CPU-only host and exports exactly what include/ghf.h declares."""
    lib = _native.load()
    names = _native.header_symbols()
    assert set(names) == set(_native.SIGNATURES), "include/ghf.h and _native.SIGNATURES disagree"
    for n in names:
        assert hasattr(lib, n), f"libghf_hip.so does not export {n}"
    assert lib.ghf_abi_version() == _native.ABI_VERSION == 15


def test_abi_argument_validation_without_a_gpu():
    """Pure host-side checks of the C ABI (no kernel is launched)."""
    lib = _native.load()
    bn, wl, cr, sc = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    ref = ctypes.byref
    assert lib.ghf_message_config(128, ref(bn), ref(wl), ref(cr), ref(sc)) == 0
    assert (bn.value, wl.value, cr.value, sc.value) == (384, 3, 76, 128)      # two fp16 pieces, block sums in registers (default)
    try:
        os.environ["GHF_KERNEL"] = "pp"
        assert lib.ghf_message_config(128, ref(bn), ref(wl), ref(cr), ref(sc)) == 0
        assert (bn.value, wl.value, cr.value, sc.value) == (216, 1, 48, 128)  # fp32 MFMA contraction
        os.environ["GHF_KERNEL"] = "hx"                                        # (a retired name — round 1's LDS-sum kernel: the exact one)
        assert lib.ghf_message_config(128, ref(bn), ref(wl), ref(cr), ref(sc)) == 0
        assert (bn.value, wl.value) == (216, 1)
    finally:
        del os.environ["GHF_KERNEL"]
    assert lib.ghf_split_rows_bytes(10, 128, 3) == 10 * 128 * 4 + 40 and lib.ghf_split_rows_bytes(10, 128, 2) == 0
    assert lib.ghf_split_rows_bytes(10, 128, 1) == 0
    assert lib.ghf_weights_bytes(4, 128, 128, 3) == 4 * 2 * 128 * 128 * 4 + 16 and lib.ghf_weights_bytes(4, 8, 24, 0) == 4 * 8 * 24 * 4
    assert lib.ghf_message_config(64, ref(bn), ref(wl), ref(cr), ref(sc)) == 0 and (bn.value, wl.value, cr.value) == (192, 3, 64)
    assert lib.ghf_message_config(20, ref(bn), ref(wl), ref(cr), ref(sc)) == 0
    assert (bn.value, wl.value, cr.value, sc.value) == (1, 0, 0, 0)
    assert lib.ghf_message_config(16, None, None, None, None) == -1
    assert b"null" in lib.ghf_last_error()
    assert lib.ghf_plan_max_chunks(1000, 5000, 7, 216, 48) >= 5000 // 48 + 1
    assert lib.ghf_plan_max_chunks(1000, 5000, 7, 1, 0) == 0
    assert lib.ghf_plan_max_items(1000, 5000, 7, 216, 48, 128) >= 5
    assert lib.ghf_input_proj_fwd(None, None, None, 4, 4, 4, None, None, 0, None) == -1
    assert lib.ghf_tail_fwd(None, None, None, None, 1e-5, 0, 1, 8, None, None, None) == -1


def test_message_config_is_safe_as_the_first_call_of_a_process():
    """message_config takes the module lock to name a kernel; loading the library takes it too (a fresh process whose first
    native call is a plan build must not deadlock)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from graph_hypernetwork_forge_amd import _native; "
            "print(_native.message_config(128)[0], _native.exact_config(128)[0], _native.exact_config(20)[0])" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.split() == ["384", "216", "1"], out.stderr[-500:]


def test_header_has_no_torch_or_cxx_types():
    with open(os.path.join(_build.INCLUDE, "ghf.h")) as f:
        text = f.read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)              # declarations only, comments stripped
    assert 'extern "C"' in code
    assert not re.search(r"torch|at::|std::|hip/hip_runtime", code)
    assert set(re.findall(r"#include\s+<([^>]+)>", code)) == {"stddef.h", "stdint.h"}


# ---- module mirror (reference tests/test_hypergnn.py, tests/test_weight_generator.py) ---------

def test_state_dict_keys_match_the_reference_names():
    for name, cfg in cases.MODELS.items():
        m = HyperGNN(cfg.text_dim, cfg.node_feat_dim, cfg.hidden_dim, cfg.num_layers, char_emb_dim=cfg.char_emb_dim)
        params = cfg.params()
        assert set(m.state_dict().keys()) == set(params.keys()), name
        m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    g = WeightGenerator(32, 16, 16, hidden_dim=64, dropout=0.25)      # Dropout shifts Sequential indices to 0,3,6
    assert "generators.W_msg.6.weight" in g.state_dict() and "generators.bias.3.bias" in g.state_dict()
    assert sorted(g.log_scales.keys()) == ["W_msg", "W_self", "bias"]
    assert g.state_dict()["log_scales.W_msg"].shape == (1,)


def test_constructor_contract():
    with pytest.raises(ValueError):
        HyperGNN(text_dim=32, node_feat_dim=8, hidden_dim=16, num_layers=0)
    for bad in ((0, 4, 4), (4, 0, 4), (4, 4, -1)):
        with pytest.raises(ValueError):
            WeightGenerator(*bad)
    for bad_p in (-0.1, 1.5):                                         # nn.Dropout / F.dropout of the reference: ValueError
        with pytest.raises(ValueError):
            HyperGNN(text_dim=32, node_feat_dim=8, hidden_dim=16, dropout=bad_p)
        with pytest.raises(ValueError):
            WeightGenerator(32, 16, 16, dropout=bad_p)
    from graph_hypernetwork_forge_amd.models.weight_generator import draw_mask
    assert float(draw_mask((4, 5), "cpu", 1.0).abs().max()) == 0.0     # p = 1 drops everything (no 0/0)
    mk = draw_mask((1000,), "cpu", 0.25)
    assert set(np.unique(mk.numpy()).tolist()) <= {0.0, np.float32(1.0 / 0.75).item()}
    m = HyperGNN(text_dim=32, node_feat_dim=8, hidden_dim=16, num_layers=3)
    assert len(m.weight_generators) == len(m.layer_norms) == m.num_layers == 3
    assert m.weight_generators[0].hidden_dim == 64 and m.num_parameters() > 0
    g = WeightGenerator(32, 16, 16, hidden_dim=256, num_hidden=0)
    assert g.generators["W_msg"][0].in_features == 32                  # single Linear when num_hidden == 0
    last = WeightGenerator(32, 16, 16).generators["W_msg"][-1]
    assert float(last.bias.abs().max()) == 0.0 and float(last.weight.std()) < 0.02
    with pytest.raises(RuntimeError):                                            # HIP only, like every other entry point
        m.score_triple(torch.ones(4, 16), torch.ones(4, 16))


def test_forward_validation_happens_before_any_device_work():
    kg = ToyKnowledgeGraph(feat_dim=16)
    m = HyperGNN(text_dim=32, node_feat_dim=16, hidden_dim=16).eval()
    with pytest.raises(ValueError):
        m(kg.node_features, kg.edge_index, kg.edge_texts[:-1])
    with pytest.raises(ValueError):
        with torch.no_grad():
            m(kg.node_features[:, :8], kg.edge_index, kg.edge_texts)
    with pytest.raises(RuntimeError, match="no CPU"):                  # CPU tensors: fail loudly, no fallback
        with torch.no_grad():
            m(kg.node_features, kg.edge_index, kg.edge_texts)
    with pytest.raises(RuntimeError, match="no CPU"):
        with torch.no_grad():
            WeightGenerator(32, 16, 16)(torch.randn(32))


def test_toy_kg_matches_the_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "toy_features.npz"))
    kg = ToyKnowledgeGraph(feat_dim=16)
    assert kg.num_nodes == 8 and kg.num_edges == 11 and len(kg.relation_types) == 7
    assert np.array_equal(kg.edge_index.numpy(), g["edge_index"]) and kg.edge_texts == g["edge_texts"].tolist()
    assert np.array_equal(kg.node_features.numpy(), g["node_features"])


def test_text_encoder_tokenisation_and_no_cpu_path():
    enc = TextEncoder(text_dim=32, char_emb_dim=16).eval()
    texts = ["knows", "", "café → 東京", "a", "is parent of"]
    assert enc._codes("") == [0] and enc._codes("é") == [127]                       # reference hypergnn.py:68-70
    ids, lens = enc._token_matrix(texts, torch.device("cpu"))
    assert ids.dtype == torch.int32 and ids.shape == (5, 12) and lens.tolist() == [5, 1, 9, 1, 12]
    assert ids[0, :5].tolist() == [ord(c) for c in "knows"] and ids[1].tolist() == [0] * 12
    assert ids[2, :9].tolist() == [99, 97, 102, 127, 32, 127, 32, 127, 127]
    assert enc._token_matrix(texts, torch.device("cpu"))[0] is ids                   # cached per list of strings
    with torch.no_grad(), pytest.raises(RuntimeError):                                # no CPU / eager fallback
        enc(texts, torch.device("cpu"))


# ---- plan host logic --------------------------------------------------------------------------------

def test_relation_ids_first_appearance_order():
    texts = ["b", "a", "b", "c", "a"]
    uniq, ids = relation_ids(texts)
    ouniq, oids = O.relation_ids(texts)
    assert uniq == ouniq == ["b", "a", "c"] and ids.tolist() == oids.tolist() == [0, 1, 0, 2, 1]
    assert ids.dtype == np.int64


def test_relation_ids_by_object_identity_equal_the_reference_mapping(monkeypatch):
    """plan.relation_ids maps long lists by object identity first (ghf_host_word_ids) and must give exactly the reference's
    value-keyed, first-appearance mapping (models/hypergnn.py:264-268) — also when distinct objects hold equal strings, for
    tuples, and when the list references more distinct objects than the fast path takes (then it falls back)."""
    from graph_hypernetwork_forge_amd import plan as plan_mod

    def reference(texts):
        unique = list(dict.fromkeys(texts))
        lut = {t: i for i, t in enumerate(unique)}
        return unique, np.array([lut[t] for t in texts], dtype=np.int64)

    names = [f"relation_{i:04d}" for i in range(7)]
    shared = [names[(5 * i * i + 3) % 7] for i in range(20_000)]                         # few objects, many references
    equal_but_distinct = ["".join(["rel_", str(i % 5)]) for i in range(20_000)]          # 20 k objects, 5 strings
    mixed = shared[:6000] + equal_but_distinct[:6000] + ["", "caf\u00e9", ""] * 10
    for texts in (shared, equal_but_distinct, mixed, tuple(shared), shared[:100]):
        want_u, want_ids = reference(texts)
        got_u, got_ids = plan_mod.relation_ids(texts)
        assert got_u == want_u and got_ids.dtype == np.int64 and np.array_equal(got_ids, want_ids)
    monkeypatch.setattr(plan_mod, "MAX_FAST_OBJECTS", 64)                                  # more objects than the table takes
    got_u, got_ids = plan_mod.relation_ids(equal_but_distinct)
    want_u, want_ids = reference(equal_but_distinct)
    assert got_u == want_u and np.array_equal(got_ids, want_ids)


def test_relation_ids_on_several_host_threads_keep_first_appearance_order():
    """ghf_host_word_ids maps the array beyond a prefix on host threads: relations that first appear late — in different
    threads' stretches, some in more than one — still get their ids by first appearance (reference models/hypergnn.py:264-268)."""
    from graph_hypernetwork_forge_amd import plan as plan_mod
    n = 1_600_000
    names = [f"relation_{i:03d}" for i in range(40)]
    rng = np.random.default_rng(5)
    pick = rng.integers(0, 8, size=n)                                       # the first eight everywhere
    for first, rel in ((70_000, 8), (400_000, 9), (400_001, 10), (650_000, 9), (900_000, 11), (1_200_000, 10), (1_599_999, 12)):
        pick[first] = rel                                                   # late-comers, some met by two threads
    pick[1_000_000:1_000_050] = np.arange(13, 38).repeat(2)                 # a burst of new ones inside one stretch
    texts = [names[i] for i in pick.tolist()]
    unique, ids, objs = plan_mod.relation_ids(texts, want_objects=True)
    want_u = list(dict.fromkeys(texts))
    lut = {t: i for i, t in enumerate(want_u)}
    assert unique == want_u and ids.dtype == np.int64
    assert np.array_equal(ids, np.array([lut[t] for t in texts], dtype=np.int64))
    assert objs is not None and all(a is b for a, b in zip(objs, want_u))
    assert plan_mod.relation_ids(texts[:100], want_objects=True)[2] is None  # short lists: the dict path, no objects


def test_cache_entries_that_hold_the_distinct_objects():
    """PlanCache.put(objects=...): the entry keeps the list's distinct string objects and the checksums of its pointer array
    (no copy of the list); an untouched list is a hit, any entry that points elsewhere — another relation, or an equal string
    in a new object — makes the entry stale (the caller maps the list again: reference models/hypergnn.py:264-268)."""
    from graph_hypernetwork_forge_amd import plan as plan_mod
    n = plan_mod.FULL_FINGERPRINT_MAX + 50_000
    names = [f"relation_{i:04d}" for i in range(16)]
    texts = [names[(7 * i) % 16] for i in range(n)]
    _, _, objs = plan_mod.relation_ids(texts, want_objects=True)
    ei, dev = torch.zeros(2, n, dtype=torch.int64), torch.device("cpu")
    PlanCache.key(ei, texts, 3, 16, dev)                                    # (draws the sampled positions of this length)
    sampled = set(plan_mod._SAMPLE_IDX[n][0])
    pos = next(i for i in range(n // 2, n) if i not in sampled)            # an edit the key cannot see
    for background in (False, True):
        cache = PlanCache()
        key = PlanCache.key(ei, texts, 3, 16, dev)
        fut = cache.put(key, "plan", ei, texts, objects=objs, background=background)
        assert (fut is not None) == background
        if fut is not None:
            fut.result()
        ent = cache._entries[key]
        assert ent[3][0] is None and ent[3][2] is objs                      # no copy of the list
        assert cache.verifier(key, texts)() and cache.get(key) == "plan"
        old = texts[pos]
        texts[pos] = "".join(["relation_", old[9:]])                        # an equal string in a new object
        assert PlanCache.key(ei, texts, 3, 16, dev) == key
        assert not cache.verifier(key, texts)() and cache.get(key) is None and cache.stale == 1
        texts[pos] = old


def test_long_relation_lists_are_verified_entry_by_entry():
    """A hit on a list longer than the key's fingerprint covers is confirmed against the snapshot the entry holds
    (plan.same_objects: the lists' item arrays compared bytewise; plan.same_relations: equal strings count as the same
    relations, reference models/hypergnn.py:264-268)."""
    from graph_hypernetwork_forge_amd import plan as plan_mod
    n = plan_mod.FULL_FINGERPRINT_MAX + 70_000
    names = [f"relation_{i:04d}" for i in range(16)]
    texts = [names[(7 * i) % 16] for i in range(n)]
    assert plan_mod.same_objects(texts, list(texts)) and plan_mod.same_objects(tuple(texts[:9]), texts[:9])
    assert plan_mod.same_objects([], []) and not plan_mod.same_objects(texts, texts[:-1])
    big = [names[i % 16] for i in range(3_000_000)]                        # (the threaded path)
    other = list(big)
    assert plan_mod.same_objects(big, other)
    other[2_999_999] = "x"
    assert not plan_mod.same_objects(big, other)
    ei = torch.zeros(2, n, dtype=torch.int64)
    dev = torch.device("cpu")
    cache = PlanCache()
    key = PlanCache.key(ei, texts, 3, 16, dev)
    cache.put(key, "plan", ei, texts)
    sampled = set(plan_mod._SAMPLE_IDX[n][0])
    pos = next(i for i in range(n // 2, n) if i not in sampled)            # an edit the key cannot see
    check = cache.verifier(key, texts)
    assert check is not None and check() and cache.get(key) == "plan"
    old = texts[pos]
    texts[pos] = "".join(["relation_", old[9:]])                          # an equal string, another object: same relations
    assert texts[pos] is not old and PlanCache.key(ei, texts, 3, 16, dev) == key
    assert cache.verifier(key, texts)() and cache.get(key) == "plan"
    assert cache.verifier(key, texts)()                                   # (the snapshot now holds the new object: fast path)
    texts[pos] = names[(int(old[9:]) + 1) % 16]                           # another relation at an unsampled position
    assert PlanCache.key(ei, texts, 3, 16, dev) == key
    hits = cache.hits
    assert cache.get(key) == "plan" and not cache.verifier(key, texts)()
    assert cache.get(key) is None and cache.stale == 1 and cache.hits == hits
    short = ["a"] * 10                                                     # lists the key covers whole carry no snapshot
    k2 = PlanCache.key(ei, short, 3, 16, dev)
    cache.put(k2, "p2", ei, short)
    assert cache.verifier(k2, short) is None


def test_plan_cache_keys():
    ei = torch.tensor([[0, 1, 2], [1, 2, 0]])
    texts = ["a", "b", "a"]
    dev = torch.device("cpu")
    k1 = PlanCache.key(ei, texts, 3, 16, dev)
    assert k1 == PlanCache.key(ei, texts, 3, 16, dev)
    assert k1 != PlanCache.key(ei, list(texts), 3, 16, dev)            # a different list object
    assert k1 != PlanCache.key(ei, texts, 4, 16, dev) and k1 != PlanCache.key(ei, texts, 3, 32, dev)
    ei.add_(0)                                                          # in-place edit bumps the version counter
    assert k1 != PlanCache.key(ei, texts, 3, 16, dev)
    texts[0] = "c"                                                      # the content is part of the key
    assert PlanCache.key(ei, texts, 3, 16, dev) != PlanCache.key(ei, ["a", "b", "a"], 3, 16, dev)
    # an in-place edit ANYWHERE in a list of up to 2^17 entries changes the key (the whole list is fingerprinted) ...
    names = [f"r{i % 7}" for i in range(100_000)]
    k2 = PlanCache.key(ei, names, 3, 16, dev)
    names[54_321] = "r_new"
    assert k2 != PlanCache.key(ei, names, 3, 16, dev)
    # ... a longer one is sampled at seeded-random positions (and both ends): same positions on every call
    from graph_hypernetwork_forge_amd import plan as plan_mod
    big = ["x"] * (plan_mod.FULL_FINGERPRINT_MAX + 5)
    k3 = PlanCache.key(ei, big, 3, 16, dev)
    assert k3 == PlanCache.key(ei, big, 3, 16, dev)
    big[plan_mod._SAMPLE_IDX[len(big)][0][7]] = "y"
    assert k3 != PlanCache.key(ei, big, 3, 16, dev)
    big[-1] = "z"
    assert len(plan_mod._SAMPLE_IDX[len(big)][0]) <= plan_mod.SAMPLED_POSITIONS + 2
    cache = PlanCache(capacity=2)
    for i in range(3):
        cache.put(("k", i), object(), ei, texts)
    assert len(cache) == 2 and cache.get(("k", 0)) is None and cache.get(("k", 2)) is not None
    assert (cache.hits, cache.misses) == (1, 1)
    os.environ["GHF_PLAN_CACHE"] = "0"                                  # every lookup a miss
    try:
        assert cache.get(("k", 2)) is None
    finally:
        del os.environ["GHF_PLAN_CACHE"]
    assert cache.get(("k", 2)) is not None


# ---- synthetic data -------------------------------------------------------------------------------------

def test_synth_is_deterministic_and_well_distributed():
    a = synth.make_kg(1000, 8000, 16, 8, seed=5)
    b = synth.make_kg(1000, 8000, 16, 8, seed=5)
    assert np.array_equal(a.edge_index, b.edge_index) and np.array_equal(a.node_features, b.node_features)
    assert a.edge_index.min() >= 0 and a.edge_index.max() < 1000 and a.rel_ids.max() < 16
    assert abs(float(a.node_features.mean())) < 0.05 and abs(float(a.node_features.std()) - 1) < 0.05
    assert synth.raw_u64(1, "x", 4).tolist() == synth.raw_u64(1, "x", 6)[:4].tolist()      # counter-based
    assert synth.raw_u64(1, "x", 3, offset=2).tolist() == synth.raw_u64(1, "x", 5)[2:].tolist()
    p = synth.make_kg(2000, 40000, 8, 4, seed=6, kind="powerlaw")
    deg = np.bincount(p.edge_index[1], minlength=2000)
    assert deg.max() > 20 * deg.mean()                                                       # a hub exists
    texts = p.edge_texts()
    assert len(texts) == 40000 and texts[0] is p.relation_texts[p.rel_ids[0]]
    # pinned words: any change of the generator would silently invalidate the golden fixtures
    assert synth.raw_u64(1003, "src", 2).tolist() == [int(x) for x in synth.raw_u64(1003, "src", 2)]
    assert cases.params_digest(cases.MODELS["small"].params()) == str(
        np.load(os.path.join(cases.GOLDEN_DIR, "g2_toy.npz"))["params_sha256"])


def test_header_is_plain_c_and_binds_from_c(tmp_path):
    """include/ghf.h compiles as C99 (-Wall -Werror) and a C program drives the library through dlopen: the boundary is a
    C ABI, not a Python extension (tests/c/abi_smoke.c; host-side entry points only, no GPU needed)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = os.path.join(os.path.dirname(__file__), "c", "abi_smoke.c")
    exe = str(tmp_path / "abi_smoke")
    subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-I", _build.INCLUDE, src, "-o", exe, "-ldl"],
                   check=True, capture_output=True, text=True)
    _native.load()                                                  # builds the library if it is missing
    out = subprocess.run([exe, _native.lib_path()], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert out.stdout.strip() == f"ok {_native.ABI_VERSION}"


def test_plan_config_routes_graphs_beyond_the_block_kernels_row_offsets_to_a_csr_plan():
    """the destination-block kernels gather rows through 32-bit byte offsets; graphs with more rows than those
    reach must get a CSR plan (relation-stationary layer / generic kernel, 64-bit row indices) instead of an EINVAL at the
    first layer — `bench.py --weak` runs N = 1 M x world rows."""
    from graph_hypernetwork_forge_amd import _native, plan
    bn, wl, _, _ = plan.plan_config(128, 10_000_000, 1_000_000)
    assert bn == 384 and wl == _native.WLAYOUT_SPLIT2H
    lim = plan.block_kernel_max_nodes(128, wl)
    assert 8_000_000 < lim < (1 << 32) // 516 and lim * 516 <= 0xFFFFF000
    assert plan.plan_config(128, 80_000_000, 8_000_000)[0] == 384           # bench.py --weak at 8 ranks
    assert plan.plan_config(128, 80_000_000, lim)[0] == 384
    assert plan.plan_config(128, 80_000_000, lim + 1) == plan.CSR_CONFIG
    assert plan.plan_config(64, 80_000_000, 17_000_000) == plan.CSR_CONFIG
    assert plan.plan_config(20, 1000, 1 << 40) == plan.CSR_CONFIG          # no block kernel anyway
    assert plan.block_kernel_max_nodes(128, _native.WLAYOUT_FRAG16) == ((1 << 32) - 4096) // 512
