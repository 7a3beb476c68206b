#!/bin/bash
# GPU box: hardware counters of one matrix kernel, one rocprofv3 --pmc pass per counter group (tools/wgrad_pmc_probe.py is the workload).
# (a counter name the device does not know aborts rocprofv3 and can leave it hanging: every pass runs under `timeout`)
# usage (through gpurun): bash tools/pmc_kernel_probe.sh wgrad|conv  ->  gpurun_out/pmc_<what>.txt
set -u
W=${1:-wgrad}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_$W
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" \
           "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS" \
           "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o r -- python3 $R/tools/wgrad_pmc_probe.py $W > $O/p$i.log 2>&1
done
cd $R
python3 - "$O" "$W" <<'PY' > $R/gpurun_out/pmc_$W.txt
import csv, glob, sys, collections
O, W = sys.argv[1], sys.argv[2]
key = 'wgrad_f32_kernel' if W == 'wgrad' else 'igemm_f32_kernel'
vals = collections.OrderedDict()
for f in sorted(glob.glob(O + '/p*/**/*counter_collection.csv', recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if key in r.get('Kernel_Name', '')]
    by = collections.defaultdict(list)
    for r in rows:
        by[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, v in by.items():
        vals[c] = sum(v[1:]) / max(1, len(v) - 1)  # skip the first (cold) launch
for c, v in vals.items():
    print('%-44s %18.0f' % (c, v))
g = vals.get
if g('SQ_INSTS_VMEM_RD') and g('SQ_INST_LEVEL_VMEM'):
    print('mean VMEM read latency (cycles)  %.0f' % (g('SQ_INST_LEVEL_VMEM') / g('SQ_INSTS_VMEM_RD')))
if g('SQ_INSTS_LDS') and g('SQ_INST_LEVEL_LDS'):
    print('mean LDS latency (cycles)        %.0f' % (g('SQ_INST_LEVEL_LDS') / g('SQ_INSTS_LDS')))
if g('SQ_BUSY_CYCLES') and g('SQ_VALU_MFMA_BUSY_CYCLES'):
    print('MFMA busy / SQ busy              %.3f' % (g('SQ_VALU_MFMA_BUSY_CYCLES') / g('SQ_BUSY_CYCLES')))
PY
cat $R/gpurun_out/pmc_$W.txt
grep -il "error\|invalid" $O/*.log | head
rm -rf $O/p*/
