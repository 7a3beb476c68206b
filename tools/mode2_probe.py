"""GPU box: the parity-class (MODE 2) launches of the step by tile configuration: ConvTranspose2d(64->64, 2x2, stride 2) forward
at 160->320 and the stride-2 3x3 data gradients of the backbone's stage transitions.  usage: python tools/mode2_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from gpu_util import L, rnd, DEV, igemm, pack
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def run(tag, x, wp, y, k, s, p, flops):
    row = []
    for tile in (0, 1, 2, 3, 4):
        if tile == 1 and y.shape[3] % 128:
            row.append('   -  ')
            continue
        for _ in range(3):
            igemm(x, wp, None, y, k, s, p, 1, tile=tile)
        ts = []
        for _ in range(10):
            e0.record(); igemm(x, wp, None, y, k, s, p, 1, tile=tile); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        row.append('%5.1f' % (flops / ts[3] / 1e9))
    print('%-46s TFLOP/s by tile [auto, 128x128, 256x64, 128x64, 64x64] = %s' % (tag, ' '.join(row)))


N = 16
x = torch.randn(N, 160, 160, 64, device=DEV)
w = rnd(64, 64, 2, 2, seed=1, scale=0.05)  # ConvTranspose2d weight [Cin, Cout, 2, 2]
y = torch.empty(N, 320, 320, 64, device=DEV)
run('ConvT 2x2 s2 64->64 160->320 (forward)', x, pack(w, 1, 2), y, 2, 2, 0, 2.0 * N * 160 * 160 * 64 * 64 * 4)
for (H, Ci, Co) in ((80, 64, 128), (40, 128, 256), (20, 256, 512)):
    dy = torch.randn(N, H, H, Co, device=DEV)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=0.05)
    dx = torch.empty(N, 2 * H, 2 * H, Ci, device=DEV)
    run('dgrad 3x3 s2 %d->%d to %dx%d' % (Co, Ci, 2 * H, 2 * H), dy, pack(w, 1, 2), dx, 3, 2, 1, 2.0 * N * H * H * Co * Ci * 9)
