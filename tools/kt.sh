#!/bin/bash
# rocprofv3 kernel stats of one command, top kernels only: tools/kt.sh <tag> <n> -- python3 <script> [args]   (run on the GPU box)
tag=$1; n=$2; shift 3
export TMPDIR=/tmp
rm -rf gpurun_out/_kt_$tag
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/_kt_$tag -o p --output-format csv -- "$@" > gpurun_out/_kt_$tag.log 2>&1
f=$(find gpurun_out/_kt_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$n" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[: int(sys.argv[2])]:
    print("%-52s calls %5s avg %9.1f us" % (r["Name"][:52], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf gpurun_out/_kt_$tag
