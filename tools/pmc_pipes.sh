#!/bin/bash
# Which pipe is busy in the message kernel: TA / TCP / TD, LDS, vector-memory issue, matrix pipe — one rocprofv3 --pmc pass
# per group (counters only, never combined with traces), per-launch means into gpurun_out/<prefix>_pipes.json
set -e
cd "$(dirname "$0")/.."
out=${1:-pipes}
export TMPDIR=/tmp
mkdir -p gpurun_out
groups=(
 "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum"
 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
 "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"
 "TD_TD_BUSY_sum TD_TC_STALL_sum"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
 "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA"
)
i=0
: > gpurun_out/${out}_pipes.csv
for g in "${groups[@]}"; do
  rm -rf gpurun_out/_pp_$i
  echo "pass $i: $g"
  timeout -k 5 150 rocprofv3 --pmc $g -d gpurun_out/_pp_$i -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --kernel-reps 3 > gpurun_out/_pp_$i.log 2>&1 || { echo "group $i failed"; grep -m1 -i "error code" gpurun_out/_pp_$i.log; }
  f=$(find gpurun_out/_pp_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && grep -E 'Counter_Name|message_bx_kernel' "$f" >> gpurun_out/${out}_pipes.csv
  i=$((i+1))
done
python3 - "$out" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
rows = [r for r in csv.reader(open(f"gpurun_out/{out}_pipes.csv"))]
hdr = None
for r in rows:
    if "Counter_Name" in r:
        hdr = r; continue
    d = dict(zip(hdr, r))
    acc[d["Counter_Name"]].append(float(d["Counter_Value"]))
res = {k: sum(v) / len(v) for k, v in acc.items()}
json.dump(res, open(f"gpurun_out/{out}_pipes.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
