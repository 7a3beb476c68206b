#!/bin/bash
# GPU box: kernel trace of bench.py in one math mode -> gpurun_out/<tag>/{stats.md,breakdown.txt}.  Usage: bash tools/trace_mode.sh <tag> <bench args...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace -o t -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-modes "$@" > $O/bench.log 2>&1
cd $R
python3 tools/rocpd_stats.py $O/trace/t_results.db $O/stats.md > /dev/null
python3 tools/step_breakdown.py $O/trace/t_results.db 6 > $O/breakdown.txt 2>&1
rm -rf $O/trace
head -32 $O/breakdown.txt
