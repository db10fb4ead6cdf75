#!/bin/bash
set -o pipefail
# (A record of what was run: the build variants these lines name — FOLD*, IDXPAD*, FOLDDRAIN, LDS_PAD ... — were diagnostic
# code in message_bx.hip that exists only at commit 9c12439; the outcome is profiles/r04_hazard_bisect.txt and DESIGN.md §3.)
mkdir -p gpurun_out
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > gpurun_out/r4d_$name.log 2>&1; echo "rc=$?"; }
for v in b64DEFER1_bxFOLDDELAY30 bxLATE0_bxFOLDDELAY30 b64DEFER1 bxLATE0; do
  run $v env GHF_VARIANT=$v timeout -k 10 400 python tools/diag_fold.py 6
done
