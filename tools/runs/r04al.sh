#!/bin/bash
python3 -m pytest tests/test_model_gpu.py -m gpu -q -x -k "pyramid_conv_on_a_winograd" 2>&1 | tail -5
P="import json,sys; d=json.loads(sys.stdin.read()); print('%.2f img/s' % d['value'], d['timing']['ms_per_step_median'])"
A="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
for i in 1 2 3; do
for l in 0 1; do
echo -n "fpn_lv0_winograd$l "; DBN_FPN_LV0_WINOGRAD=$l python3 bench.py $A 2>/dev/null | tail -1 | python3 -c "$P"
done; done
