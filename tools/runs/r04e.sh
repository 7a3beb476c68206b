#!/bin/bash
O=gpurun_out/r04e; mkdir -p $O
STAGGERS=0,250,500,1000 python tools/patch_f32_probe.py > $O/patch_probe.txt 2>&1; cat $O/patch_probe.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
for i in 1 2; do
  DBN_STAGGER=0 DBN_PATCH_F32=0 $B 2>/dev/null > $O/ab_s0_p0_$i.json
  DBN_STAGGER=1000 DBN_PATCH_F32=0 $B 2>/dev/null > $O/ab_s1000_p0_$i.json
  DBN_STAGGER=500 DBN_PATCH_F32=0 $B 2>/dev/null > $O/ab_s500_p0_$i.json
  DBN_STAGGER=1000 DBN_PATCH_F32=1 $B 2>/dev/null > $O/ab_s1000_p1_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04e/ab_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'roofline', d['roofline']['kernel'], d['roofline']['frac'], 'serial', d['roofline_serial']['frac'])
    except Exception as e:
        print(f, 'ERR', e)
PY
