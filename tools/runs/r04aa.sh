#!/bin/bash
# where does the 98 ms first timed step come from?  $1 = sequence of DBN_WINOGRAD_WGRAD values, one bench process each
for w in $@; do
  DBN_WINOGRAD_WGRAD=$w python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-modes --serial-steps 1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('wwg$w %.2f img/s' % d['value'], d['timing']['ms_per_step_in_order'])"
done
