"""Compile csrc/*.hip for gfx950 into libghf_hip.so, in-tree (hipcc cross-compiles without a GPU)."""

from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from typing import List

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
# GHF_VARIANT=stamps selects the diagnostic build (-DGHF_STAMPS, in-kernel s_memtime stamps), GHF_VARIANT=ablate the
# one whose message kernels honour GHF_DEBUG_FLAGS (-DGHF_ABLATE: pieces of work switched off, wrong results);
# neither is ever the product
VARIANT = os.environ.get("GHF_VARIANT", "")
OBJ_DIR = os.path.join(CSRC, "_obj" + ("_" + VARIANT if VARIANT else ""))
LIB_PATH = os.path.join(PKG_DIR, "libghf_hip" + ("_" + VARIANT if VARIANT else "") + ".so")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")

SOURCES = ["capi.hip", "plan.hip", "text_encoder.hip", "score.hip", "backward.hip", "weightgen.hip", "weightgen_bwd.hip", "input_proj.hip", "message_generic.hip", "message_pp.hip", "message_bx.hip", "message_rs.hip", "exchange.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(INCLUDE, "ghf.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"]
import re as _re
if "stamps" in VARIANT:
    FLAGS.append("-DGHF_STAMPS")
if VARIANT == "ablate":
    FLAGS.append("-DGHF_ABLATE")
_b = _re.search(r"baux(\d+)", VARIANT)
if _b:
    FLAGS.append("-DGHF_B_AUX=" + _b.group(1))
_x = _re.search(r"bxexp(\d+)", VARIANT)
if _x:
    FLAGS.append("-DGHF_BXEXP=" + _x.group(1))         # compile-time ablations of message_bx.hip (timing only)
for _k in ("NPW", "CR", "LATE", "DEFER", "IDXWAIT", "TAILNT", "SRCNT", "PRE0"):               # message_bx.hip geometry / protocol: e.g. GHF_VARIANT=bxNPW80_bxCR64
    _g = _re.search(r"bx%s(\d+)" % _k, VARIANT)
    if _g:
        FLAGS.append("-DGHF_BX_%s=%s" % (_k, _g.group(1)))
for _k in ("NPW", "CR", "DEFER"):                         # message_bx.hip, hidden 64: e.g. GHF_VARIANT=b64NPW96_b64CR128
    _g = _re.search(r"b64%s(\d+)" % _k, VARIANT)
    if _g:
        FLAGS.append("-DGHF_BX64_%s=%s" % (_k, _g.group(1)))
_r = _re.search(r"rsexp(\d+)", VARIANT)
if _r:
    FLAGS.append("-DGHF_RSEXP=" + _r.group(1))          # message_rs.hip pass-1 ablations (timing only)
_i = _re.search(r"ipexp(\d+)", VARIANT)
if _i:
    FLAGS.append("-DGHF_IPEXP=" + _i.group(1))          # input_proj.hip timing experiments
_e = _re.search(r"eoNS(\d)(\d)", VARIANT)
if _e:                                                   # backward.hip: edge_outer_h's register sets of source / destination rows in flight
    FLAGS += ["-DGHF_EO_STAGES_A=" + _e.group(1), "-DGHF_EO_STAGES_B=" + _e.group(2)]
_e = _re.search(r"wgHU(\d+)", VARIANT)
if _e:
    FLAGS.append("-DGHF_WG_HU=" + _e.group(1))          # weightgen.hip: output units per wave and batch of wg_hidden_kernel
_e = _re.search(r"eoSRCLAST(\d)", VARIANT)
if _e:
    FLAGS.append("-DGHF_EO_SRC_LAST=" + _e.group(1))    # backward.hip: edge_outer_h's source rows requested and cut last
_e = _re.search(r"eoexp(\d+)", VARIANT)
if _e:
    FLAGS.append("-DGHF_EOEXP=" + _e.group(1))          # backward.hip: edge_outer_h ablations (timing only)
if "eopin" in VARIANT:
    FLAGS.append("-DGHF_EO_PIN")
if "eoslow" in VARIANT:
    FLAGS.append("-DGHF_EO_SLOW_FRAG")                  # debug: edge_outer_h fragments read element by element

def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libghf_hip.so")


def _stale(target: str, deps: List[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Build (if stale) and return the path of libghf_hip.so."""
    hipcc = hipcc_path()
    os.makedirs(OBJ_DIR, exist_ok=True)
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{r.stdout}\n{r.stderr}")
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ_DIR, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build(verbose=True))
