#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() { name=$1; shift; echo "== $name: $*"; ( "$@" ) > gpurun_out/r4e_$name.log 2>&1; echo "rc=$?"; }
for v in b64DEFER1 bxLATE0; do
  run $v env GHF_VARIANT=$v timeout -k 10 500 python tools/diag_rows.py 5
done
