#!/bin/bash
python3 -m pytest tests/test_ops_gpu.py -q -x -k "pack or 16bit or storage" 2>&1 | tail -2
python3 -m pytest tests/test_model_gpu.py -q -x -k "golden or fp64 or determin" 2>&1 | tail -2
P="import json,sys; d=json.loads(sys.stdin.read()); print('%.2f img/s' % d['value'], d['timing']['ms_per_step_median'])"
A="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
for i in 1 2 3; do python3 bench.py $A --math bf16 2>/dev/null | tail -1 | python3 -c "$P"; done
python3 bench.py $A 2>/dev/null | tail -1 | python3 -c "$P"
