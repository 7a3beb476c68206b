"""GPU box: one 3x3 / stride-1 convolution (forward, mode 0, and data gradient, mode 1) per backbone stage shape and tile
configuration (1 = 128x128, 2 = 256x64, 3 = 128x64, 4 = 64x64; 0 = the library's own choice), exact-fp32 path, HIP-event timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from gpu_util import L, rnd, DEV, igemm, pack
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for (N, H, C) in ((16, 160, 64), (16, 80, 128), (16, 40, 256), (16, 20, 512)):
    x = torch.randn(N, H, H, C, device=DEV)
    w = rnd(C, C, 3, 3, seed=1, scale=0.05)
    y = torch.empty(N, H, H, C, device=DEV)
    flops = 2.0 * N * H * H * C * C * 9
    for mode in (0, 1):
        wp = pack(w, mode)
        row = []
        for tile in (0, 1, 2, 3, 4):
            if tile == 1 and C % 128:
                row.append('   -  ')
                continue
            for _ in range(3):
                igemm(x, wp, None, y, 3, 1, 1, mode, tile=tile)
            ts = []
            for _ in range(10):
                e0.record(); igemm(x, wp, None, y, 3, 1, 1, mode, tile=tile); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            row.append('%5.1f' % (flops / ts[3] / 1e9))
        print('%dx%d C=%d mode %d: TFLOP/s by tile [auto, 128x128, 256x64, 128x64, 64x64] = %s   (auto picks %d)' % (
            H, H, C, mode, ' '.join(row), L().dbn_igemm_tile_config(N * H * H, C)))
