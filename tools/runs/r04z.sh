#!/bin/bash
# exact fp32: Winograd weight gradients on / off (interleaved), then the model tests that judge the gradients
A="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
for i in 1 2 3; do
  for w in 1 0; do
    DBN_WINOGRAD_WGRAD=$w python3 bench.py $A 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('wwg$w %.2f img/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"
  done
done
python3 -m pytest tests/test_model_gpu.py -q -x -k "fp64 or golden or determin or identical or graph" 2>&1 | tail -8
