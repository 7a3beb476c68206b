#!/bin/bash
# bf16 step breakdown: round-3 tree vs working tree with late weight gradients on / off (why is late0 slower than the round-3 tree?)
O=$GRAFT_REPO_ROOT/gpurun_out/r04w; mkdir -p $O
R=$GRAFT_REPO_ROOT
A="--steps 10 --warmup 3 --no-cpu-baseline --no-alt-modes --math bf16"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/t_base -o r01 -- python3 $R/.ab_base/bench.py $A > $O/base.log 2>&1
export DBN_LATE_WGRAD=0
rocprofv3 --kernel-trace -d $O/t_late0 -o r01 -- python3 $R/bench.py $A > $O/late0.log 2>&1
export DBN_LATE_WGRAD=1
rocprofv3 --kernel-trace -d $O/t_late1 -o r01 -- python3 $R/bench.py $A > $O/late1.log 2>&1
cd $R
for c in base late0 late1; do
  python3 tools/step_breakdown.py $O/t_$c/r01_results.db 6 > $O/breakdown_$c.txt 2>&1
  python3 tools/dump_step.py $O/t_$c/r01_results.db 6 > $O/dump_$c.txt 2>&1
  head -3 $O/breakdown_$c.txt; tail -1 $O/$c.log | cut -c1-200
  rm -rf $O/t_$c
done
