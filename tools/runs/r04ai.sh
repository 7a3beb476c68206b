#!/bin/bash
P="import json,sys; d=json.loads(sys.stdin.read()); print('%.2f img/s' % d['value'], d['timing']['ms_per_step_median'])"
A="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
for i in 1 2; do
for k in 256 128 192 384 512; do
echo -n "wgs$k "; DBN_WWG_WGS=$k python3 bench.py $A 2>/dev/null | tail -1 | python3 -c "$P"
done; done
