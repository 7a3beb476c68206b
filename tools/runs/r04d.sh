#!/bin/bash
O=gpurun_out/r04d; mkdir -p $O
python tools/patch_f32_probe.py > $O/patch_probe.txt 2>&1; cat $O/patch_probe.txt
python -m pytest tests/test_ops_gpu.py tests/test_loss_gpu.py -m gpu -q -x --timeout=1200 -k "pixel_patch or loss or bn_sums or batchnorm_backward_sums or conv" > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -5 $O/gpu_tests.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
for i in 1 2 3; do
  $B 2>/dev/null > $O/ab_patch1_$i.json
  DBN_PATCH_F32=0 $B 2>/dev/null > $O/ab_patch0_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04d/ab_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        h=d.get('roofline_hbm') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'roofline', d['roofline']['kernel'], d['roofline']['frac'], 'serial', d['roofline_serial']['frac'], 'loss fwd ms', (h.get('per_kernel') or {}).get('db_loss_fwd_kernel'))
    except Exception as e:
        print(f, 'ERR', e)
PY
