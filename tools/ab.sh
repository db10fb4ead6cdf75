#!/bin/bash
# A/B timing of build variants on ONE box (box-to-box spread is ~3 %):  tools/ab.sh "" opt1 "" opt1
# each argument is a GHF_VARIANT ("" = the product build); variants must have been built (GHF_VARIANT=x python -m ... _build)
for v in "$@"; do
  GHF_VARIANT=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --kernel-reps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('variant=%-8s kernel_ms=%.3f step_ms=%.2f' % ('$v' or 'product', r['ms_per_launch'], d['ms_per_step']))"
done
