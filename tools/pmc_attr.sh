#!/bin/bash
# Round 3: who causes message_bx_kernel's HBM-side traffic?  TCC request / hit / miss and FETCH_SIZE passes on the product
# build and on three compile-time ablations (wrong results, counters only): bxexp128 = one relation's weights for every
# chunk, bxexp2 = no A-tile DMA (no row gathers), bxexp1 = no weight refills.  Run on the GPU box; variants must be built.
#   -> gpurun_out/r03_attr_<variant>_{tcc,fetch}.csv and gpurun_out/r03_attr.json (per-launch means)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
kern='message_bx_kernel'
for v in product bxexp128 bxexp2 bxexp1; do
  [ "$v" = product ] && unset GHF_VARIANT || export GHF_VARIANT=$v
  for pass in tcc fetch write; do
    [ "$pass" = write ] && [ "$v" != product ] && continue
    case $pass in tcc) ctr="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum";; fetch) ctr="FETCH_SIZE";; write) ctr="WRITE_SIZE";; esac
    rm -rf gpurun_out/_attr_${v}_$pass
    timeout -k 10 300 rocprofv3 --pmc $ctr -d gpurun_out/_attr_${v}_$pass -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --kernel-reps 3 > gpurun_out/_attr_${v}_$pass.log 2>&1
    f=$(find gpurun_out/_attr_${v}_$pass -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && grep -E "Counter_Name|$kern" "$f" > gpurun_out/r03_attr_${v}_$pass.csv
    rm -rf gpurun_out/_attr_${v}_$pass
    echo "pass $v $pass done"
  done
done
python3 - <<'PY'
import csv, json, collections, glob, os
res = {}
for f in sorted(glob.glob("gpurun_out/r03_attr_*_*.csv")):
    v = os.path.basename(f)[len("r03_attr_"):].rsplit("_", 1)[0]
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    res.setdefault(v, {}).update({c: sum(x) / len(x) for c, x in acc.items()})
for v, c in res.items():
    if "FETCH_SIZE" in c:
        c["fetch_GB_raw"] = c["FETCH_SIZE"] * 1024 / 1e9
    if "TCC_MISS_sum" in c:
        c["miss_lines_GB"] = c["TCC_MISS_sum"] * 128 / 1e9
        c["req_GB"] = c["TCC_REQ_sum"] * 128 / 1e9
json.dump(res, open("gpurun_out/r03_attr.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
