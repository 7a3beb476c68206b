#!/bin/bash
# GPU box: VERDICT r5 item 6 — the bit-reproducibility probe (tools/probes/repro: fixed inputs, N launches per case, every output hashed)
# beside a SECOND PROCESS that trains on the same GPU (round 3 saw dbn_head_tail_bwd differ in ~20 of 600 launches beside a bf16 training
# process; a second STREAM of the same process never reproduced it).  usage: tools/cotenancy.sh [launches=600] [bf16|f32|none]
N=${1:-600}; MATH=${2:-bf16}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
BG=
if [ "$MATH" != none ]; then
  python3 tools/cfg_timing.py resnet18 16 640 $MATH 100000 > /tmp/cot_bg.log 2>&1 &
  BG=$!
  # wait until the companion is inside its training loop (its GPU memory is allocated and the device is busy)
  for i in $(seq 1 120); do
    busy=$(rocm-smi --showuse 2>/dev/null | grep -o 'GPU use (%): [0-9]*' | head -1 | grep -o '[0-9]*$')
    if [ "${busy:-0}" -ge 50 ]; then break; fi
    sleep 1
  done
  echo "companion process $BG ($MATH training) running, GPU use ${busy:-?} %"
fi
tools/probes/repro $N 0
RC=$?
if [ -n "$BG" ]; then kill $BG; wait $BG 2>/dev/null; fi
echo "repro exit code $RC"
