#!/bin/bash
# A/B on ONE box between two trees (default: .ab_head, a worktree of an earlier commit with its own library, and the working tree).
# Usage (through gpurun): bash tools/ab2.sh [rounds] [extra bench args, e.g. --math bf16]
R=${1:-2}; shift
A="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes $*"
for i in $(seq $R); do
  for t in ${AB_BASE:-.ab_head} .; do
    python3 $t/bench.py $A 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-9s %.2f img/s  %.3f ms/step  roofline %s %.3f' % ('$t', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac']))"
  done
done
