set -x
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
tools/pmc.sh r04_c5_kernel 'edge_transform|segment_tail|segment_partial|run_rows|split2h_rows|rs_w' -- python3 bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline --kernel-reps 2 > /dev/null && cp gpurun_out/r04_c5_kernel_pmc.json gpurun_out/r04/
cp gpurun_out/r04/r04_c5_kernel_pmc.json profiles/
name=bench_c5
rm -rf gpurun_out/_kt_$name
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/_kt_$name -o p --output-format csv -- python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --kernel-reps 2 > gpurun_out/_kt_$name.log 2>&1
f=$(find gpurun_out/_kt_$name -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/r04/r04_${name}_kernel_stats.csv
line=$(grep -E "^\{" gpurun_out/_kt_$name.log | tail -1)
[ -n "$line" ] && echo "$line" > gpurun_out/r04/r04_${name}_profiled.json
rm -rf gpurun_out/_kt_$name
python3 bench.py --workload c5 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r04/r04_bench_c5.json 2> gpurun_out/r04/bench_c5.err
tail -c 600 gpurun_out/r04/r04_bench_c5.json
