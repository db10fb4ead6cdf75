#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
run() { name=$1; shift; rm -rf gpurun_out/_p_$name; rocprofv3 --pmc "$@" -d gpurun_out/_p_$name -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --kernel-reps 3 > gpurun_out/_p_$name.log 2>&1; f=$(find gpurun_out/_p_$name -name '*counter_collection.csv' | head -1); python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'message_hx' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(f"{k:32s} {sum(v)/len(v):.4g}")
PY
}
run a SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU
run c SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES
run d SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
