#!/bin/bash
O=gpurun_out/r04p; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout=600 -k "winograd" > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -8 $O/gpu_tests.log | cut -c1-250
python -m pytest tests/test_model_gpu.py tests/test_loss_gpu.py -m gpu -q --timeout=1200 > $O/gpu_tests_model.log 2>&1; echo "pytest rc $?"; tail -8 $O/gpu_tests_model.log | cut -c1-300
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes"
for i in 1 2; do
  $B 2>/dev/null > $O/ab_wino1_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04p/ab_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], 'roofline', d['roofline']['kernel'], d['roofline']['frac'], 'serial', d['roofline_serial']['frac'])
    for k in d['kernels'][:10]: print('   ', k['kernel'][:60], k['launches'], k['ms_per_step'], k.get('frac'))
PY
