#!/bin/bash
O=gpurun_out/r04l; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout=600 -k "winograd" > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -15 $O/gpu_tests.log
python tools/winograd_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/winograd_probe.txt
