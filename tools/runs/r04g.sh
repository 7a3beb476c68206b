#!/bin/bash
O=gpurun_out/r04g; mkdir -p $O
export DBN_LIB_PATH=$PWD/db_text_minimal_amd/libdbnet_hip_trace.so
python tools/trace_probe.py 0 0 > $O/trace_gather.txt 2>&1
python tools/trace_probe.py 1 0 > $O/trace_patch.txt 2>&1
python tools/trace_probe.py 1 1000 > $O/trace_patch_stagger.txt 2>&1
cat $O/trace_gather.txt $O/trace_patch.txt $O/trace_patch_stagger.txt | grep -v amdgpu.ids | cut -c1-400
