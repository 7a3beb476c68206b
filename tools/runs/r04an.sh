#!/bin/bash
# vector-instruction diet of the Winograd kernels: tree 4d61360 (.ab_base) vs the working tree, interleaved on one box
P="import json,sys; d=json.loads(sys.stdin.read()); print('%.2f img/s' % d['value'], d['timing']['ms_per_step_median'])"
A="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
for i in 1 2 3; do
for t in .ab_base .; do echo -n "$t "; python3 $t/bench.py $A 2>/dev/null | tail -1 | python3 -c "$P"; done; done
