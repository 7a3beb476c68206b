#!/bin/bash
# GPU box: tools/cotenancy_diff.py for the product library and A/B flavours of head_loss.hip, plus the register-state probe, each beside a
# bf16 training process.  usage: tools/cotenancy_ab.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for f in "" htnodpp htw4 htw2; do
  if [ -z "$f" ]; then lib=db_text_minimal_amd/libdbnet_hip.so; else lib=db_text_minimal_amd/libdbnet_hip_$f.so; fi
  echo "== ${f:-product}"
  DBN_LIB_PATH=$R/$lib timeout 300 python3 tools/cotenancy_diff.py f32 40 bf16 2>&1 | grep -v amdgpu | grep -E "dxb|dwb|sums|ws "
done
echo "== register-state probe beside a bf16 training process"
python3 tools/cfg_timing.py resnet18 16 640 bf16 100000 > /dev/null 2>&1 &
BG=$!
sleep 25
tools/probes/cwsr_probe 150 3000
kill $BG; wait $BG 2>/dev/null
echo "== register-state probe alone"
tools/probes/cwsr_probe 50 3000
