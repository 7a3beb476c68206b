#!/bin/bash
O=gpurun_out/r04s; mkdir -p $O
python -m pytest tests -m gpu -q --timeout=1200 > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -5 $O/gpu_tests.log | cut -c1-300
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
for i in 1 2; do $B 2>/dev/null > $O/bench_$i.json; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04s/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'])
PY
