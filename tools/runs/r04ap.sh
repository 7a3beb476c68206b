#!/bin/bash
# faster batched weight pack: tree before (.ab_base) vs the working tree, interleaved on one box; fp32 and bf16
P="import json,sys; d=json.loads(sys.stdin.read()); print('%.2f img/s' % d['value'], d['timing']['ms_per_step_median'])"
A="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
for m in "" "--math bf16"; do
for i in 1 2 3; do
for t in .ab_base .; do echo -n "$t $m "; python3 $t/bench.py $A $m 2>/dev/null | tail -1 | python3 -c "$P"; done; done; done
