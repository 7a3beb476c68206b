#!/bin/bash
# A/B on ONE box: the round-1 tree (.ab_base, a git worktree of c4a1ec0 with its own libdbnet_hip.so) against the working tree,
# interleaved.  Usage (through gpurun): bash tools/ab.sh [rounds] [extra bench args]
R=${1:-2}; shift
A="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes $*"
for i in $(seq $R); do
  for t in .ab_base .; do
    python3 $t/bench.py $A 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-9s %.2f img/s  %.3f ms/step  roofline %s %.3f' % ('$t', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac']))"
  done
done
