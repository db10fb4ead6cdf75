#!/bin/bash
# Rehearsal matrix of bench.py's multi-rank branch on one card (gloo; see bench_two_ranks_one_gpu.sh): every partition / exchange
# runs end to end and prints its dist summary.  The timings mean nothing.
cd "$(dirname "$0")/.."
rc=0
while read -r args; do
  MASTER_PORT=$((29600 + RANDOM % 300)) tools/bench_two_ranks_one_gpu.sh $args > gpurun_out/_two.json 2> gpurun_out/_two.err || { echo "FAILED: $args"; tail -5 gpurun_out/_two.err; rc=1; continue; }
  python3 - "$args" <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/_two.json") if l.startswith("{")][-1]); x = d["dist"]
print(sys.argv[1], "->", x["mode"], x["exchange"], x["exchange_candidates_ms"], "ms/step", round(d["ms_per_step"], 3),
      "rows needed/received", [(r["rows_needed_per_layer"], r["rows_received_per_layer"]) for r in x["per_rank"]])
PY
done <<'LIST'
c2 --exchange auto
c3 --exchange sparse
c2 --dist-mode edges
c2 --balance edges
c2 --exchange sparse --balance edges
LIST
exit $rc
