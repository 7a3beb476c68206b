#!/bin/bash
O=gpurun_out/r04u; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
for i in 1 2 3; do
  DBN_LATE_WGRAD=1 $B 2>/dev/null > $O/ab_late1_$i.json
  DBN_LATE_WGRAD=0 $B 2>/dev/null > $O/ab_late0_$i.json
done
for i in 1 2; do
  DBN_LATE_WGRAD=1 $B --math bf16 2>/dev/null > $O/ab_bf16_late1_$i.json
  DBN_LATE_WGRAD=0 $B --math bf16 2>/dev/null > $O/ab_bf16_late0_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04u/ab_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['final_total_loss'])
    except Exception as e: print(f, 'ERR', e)
PY
python -m pytest tests/test_model_gpu.py -m gpu -q -x --timeout=1200 -k "graph or determin or reproduc or rccl or grouped or golden" 2>&1 | tail -3
