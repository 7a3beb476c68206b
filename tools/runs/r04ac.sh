#!/bin/bash
P="import json,sys; d=json.loads(sys.stdin.read()); print('%.2f img/s' % d['value'], d['timing']['ms_per_step_in_order'][:6], d['timing']['host_enqueue_ms_in_order'])"
A="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --serial-steps 1"
python3 bench.py $A 2>/dev/null | tail -1 | python3 -c "$P"
python3 bench.py $A --math bf16 2>/dev/null | tail -1 | python3 -c "$P"
python3 bench.py $A --math bf16 --graph 2>/dev/null | tail -1 | python3 -c "$P"
python3 bench.py $A --graph 2>/dev/null | tail -1 | python3 -c "$P"
python3 bench.py $A --math bf16 2>/dev/null | tail -1 | python3 -c "$P"
python3 bench.py $A --math bf16 --graph 2>/dev/null | tail -1 | python3 -c "$P"
