#!/bin/bash
# GPU box: kernel trace + two PMC passes + the plain bench line of one round.  Usage (through gpurun):
#   gpurun -- 'bash tools/profile_round.sh r02'
# Writes gpurun_out/<tag>/: kernel_stats.md (rocprofv3 --kernel-trace of bench.py), step_breakdown.txt, pmc_traffic.json (two PMC
# passes), bench_under_rocprof.json, bench_n1.json, kernel stats + bench lines for --math bf16 and bf16x3; copy the summaries into profiles/.
set -u
# fastest_step <db>: of the timed steps 5..9 the one with the shortest span (under the tracer a single step can be host-paced — the 16-bit
# step is about as long as its own enqueue — and a dump of that one would show idle gaps that the un-profiled run does not have)
fastest_step() {
  local best=6 bestus=999999999
  for k in 5 6 7 8 9; do
    us=$(python3 tools/dump_step.py $1 $k 2>/dev/null | head -1 | sed -n 's/^step [0-9]*: \([0-9]*\)\..*/\1/p')
    if [ -n "$us" ] && [ "$us" -lt "$bestus" ]; then bestus=$us; best=$k; fi
  done
  echo $best
}
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--no-cpu-baseline --no-alt-modes --no-parity"   # (the profiled passes skip the parity gate: its extra step would sit in the traces; the plain bench line below runs it)
rocprofv3 --kernel-trace -d $O/trace -o r01 -- python3 $R/bench.py --steps 10 --warmup 3 $A > $O/bench_under_rocprof.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o r01 -- python3 $R/bench.py --steps 2 --warmup 1 $A > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o r01 -- python3 $R/bench.py --steps 2 --warmup 1 $A > $O/pmc_write.log 2>&1
cd $R
python3 tools/rocpd_stats.py $O/trace/r01_results.db $O/kernel_stats.md > /dev/null
K32=$(fastest_step $O/trace/r01_results.db)
python3 tools/step_breakdown.py $O/trace/r01_results.db $K32 > $O/step_breakdown.txt 2>&1
python3 tools/pmc_traffic.py $O/pmc_fetch/r01_results.db $O/pmc_write/r01_results.db $O/pmc_traffic.json > /dev/null
grep '^{"metric"' $O/bench_under_rocprof.log > $O/bench_under_rocprof.json
cp $O/pmc_traffic.json profiles/${TAG}_pmc_traffic.json   # bench.py reads the per-kernel traffic of the newest round from profiles/
python3 bench.py --steps 20 --warmup 5 > $O/bench_n1.log 2>&1
grep '^{"metric"' $O/bench_n1.log > $O/bench_n1.json
cd /tmp
rocprofv3 --kernel-trace -d $O/trace_bf16 -o r01 -- python3 $R/bench.py --steps 10 --warmup 3 $A --math bf16 > $O/bench_bf16_under_rocprof.log 2>&1
cd $R
python3 tools/rocpd_stats.py $O/trace_bf16/r01_results.db $O/kernel_stats_bf16.md > /dev/null
grep '^{"metric"' $O/bench_bf16_under_rocprof.log > $O/bench_bf16_under_rocprof.json
cd /tmp
rocprofv3 --kernel-trace -d $O/trace_x3 -o r01 -- python3 $R/bench.py --steps 10 --warmup 3 $A --math bf16x3 > $O/bench_bf16x3_under_rocprof.log 2>&1
cd $R
python3 tools/rocpd_stats.py $O/trace_x3/r01_results.db $O/kernel_stats_bf16x3.md > /dev/null
grep '^{"metric"' $O/bench_bf16x3_under_rocprof.log > $O/bench_bf16x3_under_rocprof.json
K16=$(fastest_step $O/trace_bf16/r01_results.db)
python3 tools/step_breakdown.py $O/trace_bf16/r01_results.db $K16 > $O/step_breakdown_bf16.txt 2>&1
# round 5: the probes behind DESIGN section 13 (alone, HIP events) and the other BASELINE configurations
python3 tools/winograd_probe.py 2>/dev/null | grep -v amdgpu > $O/winograd_probe.txt
python3 tools/winograd_wgrad_probe.py 2>/dev/null | grep -v amdgpu > $O/winograd_wgrad_probe.txt
if [ -f db_text_minimal_amd/libdbnet_hip_trace.so ]; then
  DBN_LIB_PATH=$R/db_text_minimal_amd/libdbnet_hip_trace.so python3 tools/wino_trace_probe.py 2>/dev/null | grep -v amdgpu > $O/wino_trace.txt
fi
{ python3 tools/cfg_timing.py resnet50 8 800 f32 10; python3 tools/cfg_timing.py resnet50 8 800 bf16 10;
  python3 tools/cfg_timing.py deformable_resnet50 8 800 f32 10; python3 tools/cfg_timing.py deformable_resnet50 8 800 bf16 10;
  DBN_TIMING_LR=0.0002 python3 tools/cfg_timing.py deformable_resnet50 8 800 f32 10; DBN_TIMING_LR=0.0002 python3 tools/cfg_timing.py deformable_resnet50 8 800 bf16 10;
  python3 tools/cfg_timing.py resnet18 32 1280 fp16 10 eval; DBN_FOLD_EVAL_BN=0 python3 tools/cfg_timing.py resnet18 32 1280 fp16 10 eval;
  python3 tools/cfg_timing.py resnet18 16 640 f32 10 eval; DBN_FOLD_EVAL_BN=0 python3 tools/cfg_timing.py resnet18 16 640 f32 10 eval; } 2>/dev/null | grep -v amdgpu > $O/other_configs.txt
cd /tmp   # the inference configuration (BASELINE configs[4]) kernel by kernel
rocprofv3 --kernel-trace -d $O/trace_eval -o r01 -- python3 $R/tools/cfg_timing.py resnet18 32 1280 fp16 10 eval > $O/eval_fp16.log 2>&1
cd $R
python3 tools/rocpd_stats.py $O/trace_eval/r01_results.db $O/kernel_stats_eval_fp16.md > /dev/null
python3 tools/dump_step.py $O/trace/r01_results.db $K32 > $O/step_dump.txt 2>&1
python3 tools/dump_step.py $O/trace_bf16/r01_results.db $K16 > $O/step_dump_bf16.txt 2>&1
# round 6: counter evidence for the 16-bit modes and for the matrix pipe.  Every counter group is its own pass (--pmc only, no trace
# domains beside it); FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950.
cd /tmp
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch16 -o r01 -- python3 $R/bench.py --steps 2 --warmup 1 $A --math bf16 > $O/pmc_fetch16.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write16 -o r01 -- python3 $R/bench.py --steps 2 --warmup 1 $A --math bf16 > $O/pmc_write16.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_eval -o r01 -- python3 $R/tools/cfg_timing.py resnet18 32 1280 fp16 3 eval > $O/pmc_fetch_eval.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_eval -o r01 -- python3 $R/tools/cfg_timing.py resnet18 32 1280 fp16 3 eval > $O/pmc_write_eval.log 2>&1
# matrix-pipe busy cycles per kernel (SQ_VALU_MFMA_BUSY_CYCLES: cycles, summed over the SIMDs) against the kernel's own cycles (GRBM_GUI_ACTIVE)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_mfma -o r01 -- python3 $R/bench.py --steps 2 --warmup 1 $A > $O/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_mfma16 -o r01 -- python3 $R/bench.py --steps 2 --warmup 1 $A --math bf16 > $O/pmc_mfma16.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_mfma_eval -o r01 -- python3 $R/tools/cfg_timing.py resnet18 32 1280 fp16 3 eval > $O/pmc_mfma_eval.log 2>&1
cd $R
python3 tools/pmc_traffic.py $O/pmc_fetch16/r01_results.db $O/pmc_write16/r01_results.db $O/pmc_traffic_bf16.json > /dev/null
python3 tools/pmc_traffic.py $O/pmc_fetch_eval/r01_results.db $O/pmc_write_eval/r01_results.db $O/pmc_traffic_fp16.json > /dev/null
cp $O/pmc_traffic_bf16.json profiles/${TAG}_pmc_traffic_bf16.json
cp $O/pmc_traffic_fp16.json profiles/${TAG}_pmc_traffic_fp16.json
# (the plain lines of the alternative modes come AFTER the mode's traffic file exists: bench.py quotes it in roofline.traffic)
for m in bf16 bf16x3; do   # the plain (un-profiled) lines of the two alternative modes, each with its own roofline and parity object
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-modes --math $m 2>/dev/null | grep '^{"metric"' > $O/bench_n1_$m.json
done
python3 tools/pmc_mfma_busy.py $O/pmc_mfma/r01_results.db > $O/mfma_busy_f32.txt 2>&1
python3 tools/pmc_mfma_busy.py $O/pmc_mfma16/r01_results.db > $O/mfma_busy_bf16.txt 2>&1
python3 tools/pmc_mfma_busy.py $O/pmc_mfma_eval/r01_results.db > $O/mfma_busy_eval_fp16.txt 2>&1
python3 tools/mfma_peak.py 2>/dev/null | grep -v amdgpu > $O/mfma_peak.txt
python3 tools/wres_probe.py 2>/dev/null | grep -v amdgpu > $O/wres_probe.txt
rm -rf $O/pmc_fetch $O/pmc_write $O/trace $O/trace_bf16 $O/trace_x3 $O/trace_eval $O/pmc_fetch16 $O/pmc_write16 $O/pmc_fetch_eval $O/pmc_write_eval $O/pmc_mfma $O/pmc_mfma16 $O/pmc_mfma_eval   # the DBs are large; the summaries above are what is kept
ls -la $O
