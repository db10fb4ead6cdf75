#!/bin/bash
# timing of the compile-time ablation builds of message_hx.hip (GHF_VARIANT=exp<mask>, see its header)
for v in "$@"; do
  GHF_VARIANT=exp$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --kernel-reps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('exp=$v kernel_ms=%.3f step_ms=%.2f' % (r['ms_per_launch'], d['ms_per_step']))"
done
