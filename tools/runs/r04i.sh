#!/bin/bash
O=gpurun_out/r04i; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout=1200 -k "conv or igemm or pixel_patch or chunk" > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -4 $O/gpu_tests.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
for i in 1 2 3; do
  DBN_PATCH_F32=0 $B 2>/dev/null > $O/ab_bufst1_p0_$i.json
  DBN_PATCH_F32=0 DBN_LIB_PATH=$PWD/db_text_minimal_amd/libdbnet_hip_nobufst.so $B 2>/dev/null > $O/ab_bufst0_p0_$i.json
  DBN_PATCH_F32=1 $B 2>/dev/null > $O/ab_bufst1_p1_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04i/ab_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'roofline', d['roofline']['kernel'], d['roofline']['frac'], 'serial', d['roofline_serial']['frac'])
    except Exception as e:
        print(f, 'ERR', e)
PY
DBN_PATCH_F32=0 python tools/launch_table.py f32 --order 2>/dev/null | grep "igemm" | head -32
