#!/bin/bash
# diagnostic: message-kernel time under ablation flags (GHF_DEBUG_FLAGS bits: 1 no gather, 2 hot B, 4 no MFMA, 8 no scatter)
for f in "$@"; do
  GHF_DEBUG_FLAGS=$f timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --kernel-reps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('flags=$f kernel_ms=%.3f step_ms=%.2f' % (r['ms_per_launch'], d['ms_per_step']))"
done
