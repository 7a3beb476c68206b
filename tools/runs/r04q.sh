#!/bin/bash
O=gpurun_out/r04q; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout=600 -k "winograd" 2>&1 | tail -2
python tools/winograd_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/winograd_probe.txt
