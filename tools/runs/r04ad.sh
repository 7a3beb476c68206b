#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04ad; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace -o r01 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-modes > $O/bench_under_rocprof.log 2>&1
cd $R
python3 tools/step_breakdown.py $O/trace/r01_results.db 6 > $O/step_breakdown.txt 2>&1
python3 tools/dump_step.py $O/trace/r01_results.db 6 > $O/dump.txt 2>&1
head -45 $O/step_breakdown.txt
rm -rf $O/trace
