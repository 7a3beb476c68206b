"""Profile by deletion of the exact-fp32 weight-gradient kernel (-DDBN_DBG bits: 1 global loads, 2 address math, 4 staging
(register transposes + ds_write)), on the layer shapes that dominate the step.  Results are wrong by construction; only the
time matters.  usage (GPU box): python tools/wgrad_deletion_probe.py"""
import ctypes, os, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from db_text_minimal_amd import _lib
dev = 'cuda'
base = _lib.lib()
shapes = [('64->64 3x3 @160  <64,192>', 16, 160, 64, 64, 3), ('256->64 3x3 @160 <64,192>', 16, 160, 256, 64, 3),
          ('128->128 3x3 @80 <128,128>', 16, 80, 128, 128, 3), ('256->256 3x3 @40 <128,128>', 16, 40, 256, 256, 3)]
libs = sorted(glob.glob(os.path.join(ROOT, 'tools', 'probes', 'dbg', 'libwdbg_*.so')), key=lambda p: int(p.split('_')[-1][:-3]))
libs.append(os.path.join(ROOT, 'db_text_minimal_amd', 'libdbnet_hip.so'))
names = {-1: 'product library', 301: 'prefetch distance 1', 302: 'prefetch distance 2', 303: 'prefetch distance 3', 304: 'prefetch distance 4', 16: 'the three taps of a row read the same pixel (L1 re-use)', 8: 'every split reads the first pixel range (L2-resident)', 201: 'sched barriers: loads | MFMA | staging', 202: 'sched barriers: loads | MFMA | last 3 MFMA + staging', 0: 'baseline', 1: '-loads', 2: '-address math', 3: '-loads,-math', 4: '-staging', 7: 'MFMA + LDS reads only'}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st = torch.cuda.current_stream().cuda_stream
for what, N, H, Ci, Co, k in shapes:
    x = torch.randn(N, H, H, Ci, device=dev)
    dy = torch.randn(N, H, H, Co, device=dev)
    g = torch.empty(Co, Ci, k, k, device=dev)
    slab = torch.empty(base.dbn_wgrad_slab_floats(N, H, H, Co, Ci, k, k), device=dev)
    flops = 2.0 * N * H * H * Co * Ci * k * k
    print(what)
    for path in libs:
        bits = int(path.split('_')[-1][:-3]) if 'wdbg' in path else -1
        l = ctypes.CDLL(path)
        f = l.dbn_wgrad_phase_t
        f.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 12 + [ctypes.c_float, ctypes.c_void_p]
        call = lambda: f(1, 0, 0, dy.data_ptr(), x.data_ptr(), slab.data_ptr(), g.data_ptr(), N, H, H, Co, H, H, Ci, Ci, k, k, 1, 1, 1.0, st)
        for _ in range(3):
            call()
        ts = []
        for _ in range(7):
            e0.record(); call(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        tf = flops / ts[2] / 1e9
        print('   %-24s %6.1f TFLOP/s  %.3f of peak' % (names.get(bits, str(bits)), tf, tf / 157.3))
