#!/bin/bash
python3 -m pytest tests/test_model_gpu.py -q -x -k "golden or determin or graph or identical or fp64" 2>&1 | tail -3
P="import json,sys; d=json.loads(sys.stdin.read()); print('%.2f img/s' % d['value'], d['timing']['ms_per_step_median'])"
A="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
for i in 1 2 3; do
for k in 0 1; do
echo -n "fpn_repack_early$k "; DBN_FPN_REPACK_EARLY=$k python3 bench.py $A 2>/dev/null | tail -1 | python3 -c "$P"
done; done
for k in 0 1; do
echo -n "bf16 fpn_repack_early$k "; DBN_FPN_REPACK_EARLY=$k python3 bench.py $A --math bf16 2>/dev/null | tail -1 | python3 -c "$P"
done
