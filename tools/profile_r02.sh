#!/bin/bash
# Round-2 profiles (run on the GPU box; copies what is to be judged into profiles/ of the merged gpurun_out):
#   kernel-trace stats of bench.py (C3, C2) and of one GPU's C5 share, PMC (HBM-side traffic, L2 requests, matrix pipe) of the
#   dominant kernels.  Counters and traces are separate rocprofv3 runs.
set -x
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
stats() {  # name, command...
  name=$1; shift
  rm -rf gpurun_out/_kt_$name
  timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/_kt_$name -o p --output-format csv -- "$@" > gpurun_out/_kt_$name.log 2>&1
  f=$(find gpurun_out/_kt_$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/r02/r02_${name}_kernel_stats.csv
  grep -E "^\{" gpurun_out/_kt_$name.log | tail -1 > gpurun_out/r02/r02_${name}.json
}
stats bench_c3 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --kernel-reps 10
tools/pmc.sh r02_message_kernel 'message_bx_kernel' > /dev/null && cp gpurun_out/r02_message_kernel_pmc.json gpurun_out/r02_message_kernel_*.csv gpurun_out/r02/
stats bench_c2 python3 bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline --kernel-reps 20
tools/pmc.sh r02_c2_kernel 'message_pp_kernel' -- python3 bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline --kernel-reps 10 > /dev/null && cp gpurun_out/r02_c2_kernel_pmc.json gpurun_out/r02/
stats c5_shard python3 tools/c5_shard_check.py
tools/pmc.sh r02_c5_shard_kernel 'edge_transform|segment_tail|segment_partial|split2h_rows|rs_w' -- python3 tools/c5_shard_check.py > /dev/null && cp gpurun_out/r02_c5_shard_kernel_pmc.json gpurun_out/r02/
tools/pmc_pipes.sh r02_message_kernel > /dev/null 2>&1 && cp gpurun_out/r02_message_kernel_pipes.json gpurun_out/r02/
ls -la gpurun_out/r02
