#!/bin/bash
set -o pipefail
# (A record of what was run: the build variants these lines name — FOLD*, IDXPAD*, FOLDDRAIN, LDS_PAD ... — were diagnostic
# code in message_bx.hip that exists only at commit 9c12439; the outcome is profiles/r04_hazard_bisect.txt and DESIGN.md §3.)
mkdir -p gpurun_out
S="500000 5000000 32 64"
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > gpurun_out/r4c_$name.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/r4c_$name.log; }
for v in b64DEFER1_bxFOLD32 b64DEFER1_bxFOLD33 b64DEFER1_bxFOLD34 bxLATE0_bxFOLD32 bxLATE0_bxFOLD33 bxLATE0_bxFOLD34 b64DEFER1_bxFOLDDELAY30 bxLATE0_bxFOLDDELAY30 b64DEFER1_bxFOLD31 bxLATE0_bxFOLD31; do
  run $v env GHF_VARIANT=$v timeout -k 10 300 python tools/stress_repro.py $S 30
done
