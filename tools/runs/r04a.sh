#!/bin/bash
# round 4, first GPU call: reproducibility probes (product + race flavour), full GPU suite, short bench
O=gpurun_out/r04a; mkdir -p $O
( time tools/probes/repro 1000 0 ) > $O/repro_plain.txt 2>&1
( time tools/probes/repro 1000 1 ) > $O/repro_stress.txt 2>&1
( time tools/probes/repro_race 200 0 ) > $O/repro_race.txt 2>&1
tail -12 $O/repro_plain.txt $O/repro_stress.txt $O/repro_race.txt
python -m pytest tests -m gpu -q -x --timeout=1200 > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -30 $O/gpu_tests.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04a/bench.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','roofline','roofline_serial','roofline_hbm','roofline_hbm_serial','data_parallel','alt_modes'):
    print(k, json.dumps(d.get(k))[:900])
PY
