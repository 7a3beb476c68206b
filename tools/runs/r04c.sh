#!/bin/bash
# round 4, third GPU call: new kernels (DBLoss last-block finalize, grouped slab reduction) under test, then interleaved A/B of the
# grouped reduction on ONE box (box-to-box spread is a few %)
O=gpurun_out/r04c; mkdir -p $O
python -m pytest tests/test_loss_gpu.py tests/test_model_gpu.py -m gpu -q -x --timeout=1200 > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -5 $O/gpu_tests.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
for i in 1 2 3; do
  DBN_DEFER_REDUCE=1 $B 2>/dev/null > $O/ab_defer1_$i.json
  DBN_DEFER_REDUCE=0 $B 2>/dev/null > $O/ab_defer0_$i.json
done
for m in bf16; do for i in 1 2; do
  DBN_DEFER_REDUCE=1 $B --math $m 2>/dev/null > $O/ab_${m}_defer1_$i.json
  DBN_DEFER_REDUCE=0 $B --math $m 2>/dev/null > $O/ab_${m}_defer0_$i.json
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04c/ab_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        h=d.get('roofline_hbm') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'roofline', d['roofline']['frac'], 'hbm', h.get('frac'), {k:v['ms'] for k,v in (h.get('per_kernel') or {}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
