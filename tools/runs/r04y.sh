#!/bin/bash
python3 -m pytest tests/test_ops_gpu.py -q -x -k "winograd_weight_gradient" 2>&1 | tail -25
