#!/bin/bash
DBN_BENCH_CPROFILE=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-modes --serial-steps 1 2> gpurun_out/cprof.txt | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f img/s' % d['value'], d['timing']['ms_per_step_in_order'], d['timing']['host_enqueue_ms_in_order'])"
grep -v "^$" gpurun_out/cprof.txt | tail -24
