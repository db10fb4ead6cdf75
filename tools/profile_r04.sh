#!/bin/bash
# Round-4 profiles (run on the GPU box; what is to be judged is copied into gpurun_out/r04/ and from there into profiles/):
#   counters: PMC passes (HBM-side traffic, L2 requests, matrix pipe; one rocprofv3 --pmc run per counter group, never with
#             traces) of the dominant kernels at C3 / C2 / C5, each json carrying a digest of the kernel sources it measured;
#             the per-pipe passes of the C3 kernel
#   traces:   rocprofv3 --kernel-trace --stats of bench.py at C3 / C2 / C5 / C1 and of a training step, then the unprofiled
#             bench lines (C3 with its CPU baseline; C2 and C1 eager and as HIP-graph replays; C5)
#   usage: tools/profile_r04.sh [counters|traces|all]
set -x
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
stats() {  # name, command...
  name=$1; shift
  rm -rf gpurun_out/_kt_$name
  timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/_kt_$name -o p --output-format csv -- "$@" > gpurun_out/_kt_$name.log 2>&1
  f=$(find gpurun_out/_kt_$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/r04/r04_${name}_kernel_stats.csv
  t=$(find gpurun_out/_kt_$name -name '*kernel_trace.csv' | head -1)
  [ -n "$t" ] && [ "$name" = bench_c3 ] && python3 tools/timeline.py "$t" > gpurun_out/r04/r04_bench_c3_timeline.txt
  [ -n "$t" ] && [ "$name" = train_c3 ] && python3 tools/timeline_train.py "$t" > gpurun_out/r04/r04_train_c3_timeline.txt
  line=$(grep -E "^\{" gpurun_out/_kt_$name.log | tail -1)
  [ -n "$line" ] && echo "$line" > gpurun_out/r04/r04_${name}_profiled.json      # (an empty line is not written: ADVICE r2)
  rm -rf gpurun_out/_kt_$name
}
if [ "${1:-all}" != traces ]; then
tools/pmc.sh r04_message_kernel 'message_bx_kernel' > /dev/null && cp gpurun_out/r04_message_kernel_pmc.json gpurun_out/r04_message_kernel_*.csv gpurun_out/r04/
tools/pmc.sh r04_c2_kernel 'message_(pp|bx)_kernel' -- python3 bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline --kernel-reps 10 > /dev/null && cp gpurun_out/r04_c2_kernel_pmc.json gpurun_out/r04/
tools/pmc.sh r04_c5_kernel 'edge_transform|segment_tail|segment_partial|run_rows|split2h_rows|rs_w' -- python3 bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline --kernel-reps 2 > /dev/null && cp gpurun_out/r04_c5_kernel_pmc.json gpurun_out/r04/
tools/pmc_pipes.sh r04_message_kernel > /dev/null 2>&1 && cp gpurun_out/r04_message_kernel_pipes.json gpurun_out/r04/
cp gpurun_out/r04/*_pmc.json profiles/          # bench.py reads profiles/r04_*_pmc.json for roofline.traffic
fi
[ "${1:-all}" = counters ] && { ls -la gpurun_out/r04; exit 0; }
stats bench_c3 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --kernel-reps 10
stats bench_c2 python3 bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline --kernel-reps 20
stats bench_c1 python3 bench.py --workload c1 --steps 50 --warmup 5 --no-cpu-baseline --kernel-reps 20
stats bench_c5 python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --kernel-reps 2
stats train_c3 python3 tools/train_bench.py --steps 5
python3 bench.py > gpurun_out/r04/r04_bench_c3.json 2> gpurun_out/r04/bench_c3.err
python3 bench.py --workload c2 --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/r04/r04_bench_c2.json 2> gpurun_out/r04/bench_c2.err
python3 bench.py --workload c2 --no-cpu-baseline --steps 50 --warmup 5 --hip-graph > gpurun_out/r04/r04_bench_c2_hipgraph.json 2>> gpurun_out/r04/bench_c2.err
python3 bench.py --workload c1 --steps 200 --warmup 20 > gpurun_out/r04/r04_bench_c1.json 2> gpurun_out/r04/bench_c1.err
python3 bench.py --workload c1 --no-cpu-baseline --steps 200 --warmup 20 --hip-graph > gpurun_out/r04/r04_bench_c1_hipgraph.json 2>> gpurun_out/r04/bench_c1.err
python3 bench.py --workload c5 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r04/r04_bench_c5.json 2> gpurun_out/r04/bench_c5.err
python3 tools/train_bench.py --steps 5 | tail -1 > gpurun_out/r04/r04_train_c3.json
python3 tools/train_bench.py --steps 5 --score edges | tail -1 > gpurun_out/r04/r04_train_c3_score_edges.json
ls -la gpurun_out/r04
