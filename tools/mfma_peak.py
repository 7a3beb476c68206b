#!/usr/bin/env python3
"""GPU: what the fp16 matrix pipe sustains on this box with non-zero operands (tools/probes/mfma_peak.hip): TFLOP/s by accumulator
chains per wave and waves per SIMD, with small-magnitude random and with zero operands (the clock follows the power budget)."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, 'gpurun_out', 'libmfma_peak.so')
if not os.path.exists(so):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-shared', '-fPIC', os.path.join(ROOT, 'tools', 'probes', 'mfma_peak.hip'), '-o', so])
L = ctypes.CDLL(so)
L.mfma_peak.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
st = torch.cuda.current_stream().cuda_stream
out = torch.empty(1 << 22, device='cuda')
for name, data in (('random fp16 ~N(0, 0.05)', (torch.randn(4096 * 8, device='cuda') * 0.05).half()), ('zeros', torch.zeros(4096 * 8, device='cuda').half())):
    for waves_per_simd in (1, 2):
        for chains in (1, 2, 4):
            wpw, wgs, iters = 4 * waves_per_simd, 256 * 4, 2000
            for _ in range(2):
                L.mfma_peak(chains, wpw, wgs, data.data_ptr(), out.data_ptr(), iters, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            L.mfma_peak(chains, wpw, wgs, data.data_ptr(), out.data_ptr(), iters, st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            fl = 2.0 * 32 * 32 * 16 * 8 * chains * iters * wpw * wgs
            print('%-24s %d wave(s)/SIMD (x4 workgroups/CU... %d waves/wg), %d chain(s): %7.1f TFLOP/s (%.2f ms)' % (name, waves_per_simd, wpw, chains, fl / ms / 1e9, ms))
