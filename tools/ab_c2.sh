#!/bin/bash
for v in "$@"; do
  GHF_VARIANT=$v timeout -k 10 200 python bench.py --workload c2 --no-cpu-baseline --steps 50 --warmup 5 --kernel-reps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('variant=%-18s kernel_ms=%.4f step_ms=%.4f' % ('$v' or 'product', r['ms_per_launch'], d['ms_per_step']))"
done
