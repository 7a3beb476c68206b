#!/bin/bash
# GPU box: kernel trace + two PMC passes + the plain bench line of one round.  Usage (through gpurun):
#   gpurun -- 'bash tools/profile_round.sh r01e'
# Writes gpurun_out/<tag>/{trace,pmc_fetch,pmc_write}/r01_results.db, bench_under_rocprof.json, bench_n1.json;
# summarise with tools/rocpd_stats.py and tools/pmc_traffic.py and copy the summaries into profiles/.
set -u
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--no-cpu-baseline --no-alt-modes"
rocprofv3 --kernel-trace -d $O/trace -o r01 -- python3 $R/bench.py --steps 10 --warmup 3 $A > $O/bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o r01 -- python3 $R/bench.py --steps 2 --warmup 1 $A > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o r01 -- python3 $R/bench.py --steps 2 --warmup 1 $A > $O/pmc_write.log 2>&1
cd $R
python3 tools/rocpd_stats.py $O/trace/r01_results.db $O/kernel_stats.md > /dev/null
python3 tools/pmc_traffic.py $O/pmc_fetch/r01_results.db $O/pmc_write/r01_results.db $O/pmc_traffic.json > /dev/null
grep '^{"metric"' $O/bench_under_rocprof.log > $O/bench_under_rocprof.json
cp $O/pmc_traffic.json profiles/r01_pmc_traffic.json   # bench.py reads the per-kernel traffic from here
python3 bench.py --steps 20 --warmup 5 > $O/bench_n1.log 2>&1
grep '^{"metric"' $O/bench_n1.log > $O/bench_n1.json
rm -rf $O/pmc_fetch $O/pmc_write   # the DBs are large; the summaries above are what is kept
ls -la $O
