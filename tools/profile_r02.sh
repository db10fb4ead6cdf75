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
# part 1 (tools/profile_r02.sh counters): bench.py reads profiles/r02_*_pmc.json for roofline.traffic, so these come first
if [ "${1:-all}" != traces ]; then
tools/pmc.sh r02_message_kernel 'message_bx_kernel' > /dev/null && cp gpurun_out/r02_message_kernel_pmc.json gpurun_out/r02_message_kernel_*.csv gpurun_out/r02/
tools/pmc.sh r02_c2_kernel 'message_(pp|bx)_kernel' -- python3 bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline --kernel-reps 10 > /dev/null && cp gpurun_out/r02_c2_kernel_pmc.json gpurun_out/r02/
tools/pmc.sh r02_c5_kernel 'edge_transform|segment_tail|segment_partial|run_rows|split2h_rows|rs_w' -- python3 bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline --kernel-reps 2 > /dev/null && cp gpurun_out/r02_c5_kernel_pmc.json gpurun_out/r02/
tools/pmc_pipes.sh r02_message_kernel > /dev/null 2>&1 && cp gpurun_out/r02_message_kernel_pipes.json gpurun_out/r02/
cp gpurun_out/r02/*_pmc.json profiles/
fi
[ "${1:-all}" = counters ] && exit 0
# part 2 (tools/profile_r02.sh traces)
stats bench_c3 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --kernel-reps 10
stats bench_c2 python3 bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline --kernel-reps 20
stats bench_c5 python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --kernel-reps 2
stats c5_shard python3 tools/c5_shard_check.py
stats train_c3 python3 tools/train_bench.py --steps 5
# the lines themselves, unprofiled (the default one with its CPU baseline)
python3 bench.py > gpurun_out/r02/r02_bench_c3.json 2> gpurun_out/r02/bench_c3.err
python3 bench.py --workload c2 --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/r02/r02_bench_c2.json 2> gpurun_out/r02/bench_c2.err
python3 bench.py --workload c5 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r02/r02_bench_c5.json 2> gpurun_out/r02/bench_c5.err
python3 tools/train_bench.py --steps 5 | tail -1 > gpurun_out/r02/r02_train_c3.json
python3 tools/train_bench.py --steps 5 --score edges | tail -1 > gpurun_out/r02/r02_train_c3_score_edges.json
ls -la gpurun_out/r02
