#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04k; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace_bf16 -o r01 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-modes --math bf16 > $O/bench_bf16_under_rocprof.log 2>&1
cd $R
python3 tools/rocpd_stats.py $O/trace_bf16/r01_results.db $O/kernel_stats_bf16.md > /dev/null
python3 tools/step_breakdown.py $O/trace_bf16/r01_results.db 6 > $O/step_breakdown_bf16.txt 2>&1
head -45 $O/step_breakdown_bf16.txt
python3 tools/launch_table.py bf16 2>/dev/null | head -60 > $O/launch_table_bf16.txt
rm -rf $O/trace_bf16
