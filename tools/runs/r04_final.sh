#!/bin/bash
# round-4 profile set: tools/profile_round.sh (kernel traces, PMC traffic passes, bench lines) + the instruction-mix PMC passes
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_round.sh r04 > /dev/null 2>&1
O=$R/gpurun_out/r04
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $C -d $O/pmc_sq -o r01 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-modes > $O/pmc_sq.log 2>&1
rocprofv3 --pmc $C -d $O/pmc_sq16 -o r01 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-modes --math bf16 > $O/pmc_sq16.log 2>&1
cd $R
python3 tools/insn_mix.py $O/pmc_sq/r01_results.db 64 > $O/insn_mix_f32.txt 2>&1
python3 tools/insn_mix.py $O/pmc_sq16/r01_results.db 32 > $O/insn_mix_bf16.txt 2>&1
rm -rf $O/pmc_sq $O/pmc_sq16
python3 tools/launch_table.py f32 2>/dev/null | head -90 > $O/launch_table_f32.txt
head -20 $O/insn_mix_f32.txt; head -3 $O/step_breakdown.txt; python3 -c "
import json
for f in ('bench_n1','bench_n1_bf16','bench_n1_bf16x3','bench_under_rocprof'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('traffic'))
"
