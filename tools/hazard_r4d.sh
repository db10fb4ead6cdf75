#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > gpurun_out/r4d_$name.log 2>&1; echo "rc=$?"; }
for v in b64DEFER1_bxFOLDDELAY30 bxLATE0_bxFOLDDELAY30 b64DEFER1 bxLATE0; do
  run $v env GHF_VARIANT=$v timeout -k 10 400 python tools/diag_fold.py 6
done
