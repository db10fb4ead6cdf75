#!/bin/bash
# Collect PMC counters of the message kernel with rocprofv3, one pass per counter group (run on the GPU box):
#   tools/pmc.sh <out-prefix>        -> gpurun_out/<out-prefix>_{tcc,fetch,write,sq}.csv + <out-prefix>_pmc.json (per-launch means)
# Counters only (--pmc): never combined with the trace domains on this pool.
set -e
cd "$(dirname "$0")/.."
out=${1:-pmc}
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # name, counters...
  name=$1; shift
  rm -rf gpurun_out/_pmc_$name
  rocprofv3 --pmc "$@" -d gpurun_out/_pmc_$name -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --kernel-reps 5 > gpurun_out/_pmc_$name.log 2>&1
  f=$(find gpurun_out/_pmc_$name -name '*counter_collection.csv' | head -1)
  grep -E 'Counter_Name|message_(hx|sx|pp|mfma)_kernel' "$f" > gpurun_out/${out}_$name.csv
}
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
python3 - "$out" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list); kern = None
for part in ("tcc", "fetch", "write", "sq"):
    for row in csv.DictReader(open(f"gpurun_out/{out}_{part}.csv")):
        acc[row["Counter_Name"]].append(float(row["Counter_Value"])); kern = row["Kernel_Name"].split("(")[0]
res = {k: sum(v) / len(v) for k, v in acc.items()}
res["_kernel"] = kern
res["_command"] = "rocprofv3 --pmc <counters> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --kernel-reps 5 (one pass per counter group; per-launch means)"
json.dump(res, open(f"gpurun_out/{out}_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
