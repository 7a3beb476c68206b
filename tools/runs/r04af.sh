#!/bin/bash
P="import json,sys; d=json.loads(sys.stdin.read()); print('%.2f img/s' % d['value'], d['timing']['ms_per_step_median'], [ (k['kernel'][:28], k['ms_per_step']) for k in d['kernels'] if 'winograd_wgrad' in k['kernel'] or k['kernel'].startswith('wgrad_reduce')])"
A="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
for i in 1 2; do
for k in 1 2 3; do
echo -n "per_cu$k "; DBN_WWG_PER_CU=$k python3 bench.py $A 2>/dev/null | tail -1 | python3 -c "$P"
done; done
