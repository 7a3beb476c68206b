"""Profile by deletion of the parity-class launches (ConvTranspose2d 2x2 stride 2 forward, 64->64 at 160 -> 320; stride-2 3x3 data
gradient 128->64 to 160x160): -DDBN_DBG bits as tools/loop_deletion_probe.py, + 64 = no output stores.  tools/probes/dbg/libdbg_<bits>.so
are built by hand.  usage (GPU box): python tools/convt_deletion_probe.py"""
import ctypes, os, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from gpu_util import pack, rnd
dev = 'cuda'
libs = sorted(glob.glob(os.path.join(ROOT, 'tools', 'probes', 'dbg', 'libdbg_*.so')), key=lambda p: int(p.split('_')[-1][:-3]))
names = {0: 'product kernel', 63: 'k-loop: MFMAs only (prologue, epilogue intact)', 64: 'no output stores', 127: 'MFMAs only, no output stores'}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st = torch.cuda.current_stream().cuda_stream
N = 16
cases = []
x = torch.randn(N, 160, 160, 64, device=dev)
w = rnd(64, 64, 2, 2, seed=1, scale=0.05)
cases.append(('ConvT 2x2 s2 64->64 160->320', x, pack(w, 1, 2), torch.empty(N, 320, 320, 64, device=dev), 2, 2, 0, 2.0 * N * 160 * 160 * 64 * 64 * 4))
dy = torch.randn(N, 80, 80, 128, device=dev)
w = rnd(128, 64, 3, 3, seed=2, scale=0.05)
cases.append(('dgrad 3x3 s2 128->64 to 160x160', dy, pack(w, 1, 2), torch.empty(N, 160, 160, 64, device=dev), 3, 2, 1, 2.0 * N * 80 * 80 * 128 * 64 * 9))
for tag, src, wp, dst, k, s, p, flops in cases:
    print(tag)
    for tile in (0, 4):
        for path in libs:
            bits = int(path.split('_')[-1][:-3])
            f = ctypes.CDLL(path).dbn_igemm_f32
            f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 14 + [ctypes.c_void_p]
            _, Hs, Ws, Cs = src.shape
            _, Hd, Wd, Cd = dst.shape
            call = lambda: f(src.data_ptr(), wp.data_ptr(), None, dst.data_ptr(), N, Hs, Ws, Cs, Hd, Wd, Cd, k, k, s, p, 1, 0, tile, st)
            for _ in range(3):
                call()
            ts = []
            for _ in range(7):
                e0.record(); call(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            print('   tile %d  %-48s %6.1f us  %6.1f TFLOP/s  %.3f of peak' % (tile, names.get(bits, str(bits)), ts[2] * 1e3, flops / ts[2] / 1e9, flops / ts[2] / 1e9 / 157.3))

# the dedicated kernel (convt_f32.hip) of the product library against the general launch
from db_text_minimal_amd import _lib
Lp = _lib.lib()
tag, src, wp, dst, k, s, p, flops = cases[0]
for on in (1, 0):
    old = Lp.dbn_set_convt_kernel(on)
    call = lambda: Lp.dbn_igemm_f32(src.data_ptr(), wp.data_ptr(), None, dst.data_ptr(), N, 160, 160, 64, 320, 320, 64, 2, 2, 2, 0, 1, 0, 0, st)
    for _ in range(3):
        call()
    ts = []
    for _ in range(9):
        e0.record(); call(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    Lp.dbn_set_convt_kernel(old)
    print('   product library, %-40s %6.1f us  %6.1f TFLOP/s  %.3f of peak' % ('convt2x2_f32_kernel' if on else 'general parity-class launch', ts[3] * 1e3, flops / ts[3] / 1e9, flops / ts[3] / 1e9 / 157.3))
