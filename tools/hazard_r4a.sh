#!/bin/bash
# Round 4, the hidden-64 hazard: first set of experiments (one gpurun call).  Logs under gpurun_out/r4a_*.log
# (A record of what was run: the build variants these lines name — FOLD*, IDXPAD*, FOLDDRAIN, LDS_PAD ... — were diagnostic
# code in message_bx.hip that exists only at commit 9c12439; the outcome is profiles/r04_hazard_bisect.txt and DESIGN.md §3.)
set -o pipefail
mkdir -p gpurun_out
S="500000 5000000 32 64"
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > gpurun_out/r4a_$name.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r4a_$name.log; }
run late0_pad   env GHF_VARIANT=bxLATE0 GHF_BX_LDS_PAD=40000 timeout -k 10 300 python tools/stress_repro.py $S 20
run attr_defer1 env GHF_VARIANT=b64DEFER1 timeout -k 10 400 python tools/diag_attr.py 5
run attr_late0  env GHF_VARIANT=bxLATE0 timeout -k 10 400 python tools/diag_attr.py 5
run fold3_defer1 env GHF_VARIANT=b64DEFER1_bxFOLD31 timeout -k 10 300 python tools/stress_repro.py $S 20
run fold3_late0  env GHF_VARIANT=bxLATE0_bxFOLD31 timeout -k 10 300 python tools/stress_repro.py $S 20
run rev_defer1  env GHF_VARIANT=b64DEFER1_bxREV1 timeout -k 10 400 python tools/diag_attr.py 4
run rev_late0   env GHF_VARIANT=bxLATE0_bxREV1 timeout -k 10 400 python tools/diag_attr.py 4
