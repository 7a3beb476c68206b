#!/bin/bash
O=gpurun_out/r04n; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout=600 -k "winograd" > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -15 $O/gpu_tests.log | cut -c1-250
python -m pytest tests/test_model_gpu.py tests/test_loss_gpu.py -m gpu -q --timeout=1200 > $O/gpu_tests_model.log 2>&1; echo "pytest rc $?"; tail -12 $O/gpu_tests_model.log | cut -c1-300
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
for i in 1 2 3; do
  DBN_WINOGRAD=1 $B 2>/dev/null > $O/ab_wino1_$i.json
  DBN_WINOGRAD=0 $B 2>/dev/null > $O/ab_wino0_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04n/ab_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'roofline', d['roofline']['kernel'], d['roofline']['frac'], 'serial', d['roofline_serial']['frac'], 'loss', d['final_total_loss'])
    except Exception as e:
        print(f, 'ERR', e)
PY
python tools/launch_table.py f32 2>/dev/null | head -40 > $O/launch_table.txt
