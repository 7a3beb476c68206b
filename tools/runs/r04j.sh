#!/bin/bash
O=gpurun_out/r04j; mkdir -p $O
python -m pytest tests -m gpu -q --timeout=1200 > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -6 $O/gpu_tests.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
for i in 1 2 3; do
  $B 2>/dev/null > $O/ab_new_$i.json
  DBN_PATCH_F32=0 DBN_LIB_PATH=$PWD/db_text_minimal_amd/libdbnet_hip_nobufst.so $B 2>/dev/null > $O/ab_old_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04j/ab_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        h=d['roofline_hbm']
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'roofline', d['roofline']['kernel'], d['roofline']['frac'], 'serial', d['roofline_serial']['frac'], 'hbm', h['frac'], h['per_kernel']['db_loss_fwd_kernel'])
    except Exception as e:
        print(f, 'ERR', e)
PY
