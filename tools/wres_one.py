#!/usr/bin/env python3
"""GPU: ONE shape of tools/wres_probe.py, a few launches (for rocprofv3 passes).  usage: wres_one.py <Cs> <Cd> <N> <H> <W> <fp16|bf16> <mode> [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import wres_probe as P
Cs, Cd, N, H, W = (int(v) for v in sys.argv[1:6])
dt = torch.float16 if sys.argv[6] == 'fp16' else torch.bfloat16
mode = int(sys.argv[7])
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 3
print('%.1f us' % P.time_conv(dt, N, H, W, Cs, Cd, mode, reps=reps))
