"""Profile by deletion: the exact-fp32 igemm kernel built with pieces of its k-loop compiled out (-DDBN_DBG bits: 1 A-panel
loads, 2 weight loads, 4 staging ds_writes, 8 LDS fragment reads, 16 barrier, 32 address math), timed on the layer shapes that
dominate the step.  Results are WRONG by construction; only the time matters.  tools/probes/dbg/libdbg_<bits>.so are built by
hand.  usage (GPU box): python tools/loop_deletion_probe.py"""
import ctypes, os, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from db_text_minimal_amd import _lib
dev = 'cuda'
base = _lib.lib()
shapes = [('64->64 3x3 @160 (128x64)', 16, 160, 64, 64, 3), ('256->64 3x3 @160 (128x64)', 16, 160, 256, 64, 3),
          ('256->256 3x3 @40 (128x128)', 16, 40, 256, 256, 3), ('128->128 3x3 @80 (128x64)', 16, 80, 128, 128, 3)]
libs = sorted(glob.glob(os.path.join(ROOT, 'tools', 'probes', 'dbg', 'libdbg_*.so')), key=lambda p: int(p.split('_')[-1][:-3]))
names = {0: 'baseline', 1: '-A loads', 2: '-B loads', 3: '-A,B loads', 4: '-ds_write', 7: '-loads,-ds_write', 8: '-ds_read', 15: '-loads,-write,-read',
         16: '-barrier', 32: '-address math', 63: 'MFMA only'}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st = torch.cuda.current_stream().cuda_stream
for what, N, H, Ci, Co, k in shapes:
    x = torch.randn(N, H, H, Ci, device=dev)
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
    y = torch.empty(N, H, H, Co, device=dev)
    wp = torch.empty(base.dbn_igemm_panel_floats(Co, Ci, k, k, 0, 1), device=dev)
    base.dbn_pack_weights(w.data_ptr(), Co, Ci, k, k, 0, 1, wp.data_ptr(), st)
    flops = 2.0 * N * H * H * Co * Ci * k * k
    row = []
    for path in libs:
        bits = int(path.split('_')[-1][:-3])
        l = ctypes.CDLL(path)
        f = l.dbn_igemm_f32
        f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 14 + [ctypes.c_void_p]
        call = lambda: f(x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, H, Ci, H, H, Co, k, k, 1, 1, 0, 0, 0, st)
        for _ in range(3):
            call()
        ts = []
        for _ in range(7):
            e0.record(); call(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        row.append((bits, flops / ts[2] / 1e9))
    print(what)
    for bits, tf in row:
        print('   %-22s %6.1f TFLOP/s  %.3f of peak' % (names.get(bits, str(bits)), tf, tf / 157.3))
