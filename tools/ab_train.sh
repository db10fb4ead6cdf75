#!/bin/bash
# same-box A/B of the C3 training step: tools/ab_train.sh "ENV=VAL ..." "ENV=VAL ..." (each argument one configuration, run twice, interleaved)
for rep in 1 2; do
  for cfg in "$@"; do
    env $cfg timeout -k 10 300 python3 tools/train_bench.py --steps 8 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-40s step %.2f ms fwd %.2f bwd %.2f' % ('$cfg', d['ms_per_train_step'], d['forward_ms'], d['backward_ms']))"
  done
done
