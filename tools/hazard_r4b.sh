#!/bin/bash
# Round 4, the hidden-64 hazard, second set: LDS reads in flight while the VGPR indexing mode is on
# (A record of what was run: the build variants these lines name — FOLD*, IDXPAD*, FOLDDRAIN, LDS_PAD ... — were diagnostic
# code in message_bx.hip that exists only at commit 9c12439; the outcome is profiles/r04_hazard_bisect.txt and DESIGN.md §3.)
set -o pipefail
mkdir -p gpurun_out
S="500000 5000000 32 64"
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > gpurun_out/r4b_$name.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/r4b_$name.log; }
run probe timeout -k 10 120 tools/micro/gpr_idx_lds_probe
run fd_defer1 env GHF_VARIANT=b64DEFER1_bxFOLDDRAIN1 timeout -k 10 300 python tools/stress_repro.py $S 40
run fd_late0  env GHF_VARIANT=bxLATE0_bxFOLDDRAIN1 timeout -k 10 300 python tools/stress_repro.py $S 40
run defer1    env GHF_VARIANT=b64DEFER1 timeout -k 10 300 python tools/stress_repro.py $S 10
run late0     env GHF_VARIANT=bxLATE0 timeout -k 10 300 python tools/stress_repro.py $S 10
