"""GPU: for every distinct conv shape of the bs16 640x640 step, time each igemm tile configuration (forward and data
gradient) and compare with the one dbn_igemm_tile_config picks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from gpu_util import L, DEV, pack, igemm
N = 16
SHAPES = [  # (Cin, Cout, k, stride, pad, H)
    (64, 64, 3, 1, 1, 160), (64, 128, 3, 2, 1, 160), (128, 128, 3, 1, 1, 80), (128, 256, 3, 2, 1, 80), (256, 256, 3, 1, 1, 40),
    (256, 512, 3, 2, 1, 40), (512, 512, 3, 1, 1, 20), (256, 64, 3, 1, 1, 160), (64, 64, 3, 1, 1, 80), (64, 64, 3, 1, 1, 40),
    (64, 128, 1, 2, 0, 160), (128, 256, 1, 2, 0, 80), (256, 512, 1, 2, 0, 40), (64, 64, 1, 1, 0, 160), (128, 64, 1, 1, 0, 80),
    (256, 64, 1, 1, 0, 40), (512, 64, 1, 1, 0, 20), (64, 256, 4, 2, 1, 160), (64, 256, 6, 4, 1, 160), (64, 256, 10, 8, 1, 160)]
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (Ci, Co, k, s, p, H) in SHAPES:
    Ho = (H + 2 * p - k) // s + 1
    x = torch.randn(N, H, H, Ci, device=DEV); y = torch.randn(N, Ho, Ho, Co, device=DEV)
    w = torch.randn(Co, Ci, k, k) * 0.05
    for mode in ('fwd', 'dgrad'):
        if mode == 'dgrad' and s > 2: continue
        wp = pack(w, 0 if mode == 'fwd' else 1, s)
        M, Cd = (N * Ho * Ho, Co) if mode == 'fwd' else (N * H * H, Ci)
        chosen = L().dbn_igemm_tile_config(M, Cd)
        if chosen == 1 and Cd % 128: chosen = 3
        res = {}
        for t in (1, 2, 3, 4):
            if t == 1 and Cd % 128: continue
            f = (lambda t=t: igemm(x, wp, None, y, k, s, p, 0, 0, t)) if mode == 'fwd' else (lambda t=t: igemm(y, wp, None, x, k, s, p, 1, 0, t))
            res[t] = timeit(f)
        best = min(res, key=res.get)
        flag = '' if res[chosen] <= 1.03 * res[best] else '   <-- chosen %.0f%% slower than best' % (100 * (res[chosen] / res[best] - 1))
        print('%-5s %3d->%3d k%d s%d @%3d  M=%7d Cd=%3d  chosen %d  %s%s' % (mode, Ci, Co, k, s, H, M, Cd, chosen, '  '.join('%d:%.3f' % (t, v) for t, v in res.items()), flag))
