#!/bin/bash
# Rehearsal of bench.py's multi-rank branch on ONE GPU: two processes share the card and exchange through gloo (RCCL
# refuses two ranks on one device).  Checks that the N > 1 code path runs end to end for every partition / exchange;
# the timing means nothing.   usage: tools/bench_two_ranks_one_gpu.sh [workload] [extra bench.py flags ...]
cd "$(dirname "$0")/.."
wl=${1:-c2}; shift
export MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29541} WORLD_SIZE=2 LOCAL_RANK=0 GHF_DIST_BACKEND=gloo
RANK=1 python bench.py --gpus 2 --workload $wl --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> gpurun_out/bench_rank1.err &
pid=$!
RANK=0 python bench.py --gpus 2 --workload $wl --steps 3 --warmup 1 --no-cpu-baseline "$@"
rc=$?
wait $pid || rc=1
exit $rc
