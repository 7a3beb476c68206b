#!/bin/bash
# bf16 on ONE box: round-3 tree (.ab_base) vs the working tree (late weight gradients on / off), interleaved
A="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1 --math bf16"
for i in 1 2 3; do
  for cfg in "base" "late1" "late0"; do
    if [ $cfg = base ]; then t=.ab_base; e=""; else t=.; e="DBN_LATE_WGRAD=${cfg#late}"; fi
    env $e python3 $t/bench.py $A 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-6s %.2f img/s  %.3f ms/step' % ('$cfg', d['value'], d['ms_per_step']))"
  done
done
