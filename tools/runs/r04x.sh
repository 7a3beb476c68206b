#!/bin/bash
# bf16, immediate weight-gradient launch: which round-4 commit made it slower than the round-3 tree?  (worktrees .ab_*)
A="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1 --math bf16"
export DBN_LATE_WGRAD=0
for i in 1 2; do
  for t in .ab_base .ab_1 .ab_2 .ab_3 .; do
    python3 $t/bench.py $A 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-9s %.2f img/s  %.3f ms/step' % ('$t', d['value'], d['ms_per_step']))"
  done
done
