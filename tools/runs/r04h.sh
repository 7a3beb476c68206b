#!/bin/bash
O=gpurun_out/r04h; mkdir -p $O
DBN_LIB_PATH=$PWD/db_text_minimal_amd/libdbnet_hip_trace.so python tools/trace_probe.py 0 0 1 > $O/trace_gather_prio.txt 2>&1
DBN_LIB_PATH=$PWD/db_text_minimal_amd/libdbnet_hip_trace.so python tools/trace_probe.py 1 0 1 > $O/trace_patch_prio.txt 2>&1
cat $O/trace_gather_prio.txt $O/trace_patch_prio.txt | grep -v "amdgpu.ids\|start times" | cut -c1-300
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
for i in 1 2 3; do
  DBN_PHASE_PRIO=0 DBN_PATCH_F32=0 $B 2>/dev/null > $O/ab_prio0_p0_$i.json
  DBN_PHASE_PRIO=1 DBN_PATCH_F32=0 $B 2>/dev/null > $O/ab_prio1_p0_$i.json
  DBN_PHASE_PRIO=1 DBN_PATCH_F32=1 $B 2>/dev/null > $O/ab_prio1_p1_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04h/ab_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'roofline', d['roofline']['kernel'], d['roofline']['frac'], 'serial', d['roofline_serial']['frac'])
    except Exception as e:
        print(f, 'ERR', e)
PY
