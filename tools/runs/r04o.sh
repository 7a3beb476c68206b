#!/bin/bash
O=gpurun_out/r04o; mkdir -p $O
python -m pytest tests -m gpu -q --timeout=1200 > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -12 $O/gpu_tests.log | cut -c1-250
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2>$O/bench.err; echo rc $?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04o/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'], d['roofline_serial'])
for k in d['kernels'][:12]: print(k['kernel'][:60], k['launches'], k['ms_per_step'], k.get('ms_min_max'), k.get('frac'))
print(d['alt_modes'])
PY
python tools/winograd_probe.py 2>&1 | grep -v amdgpu.ids
