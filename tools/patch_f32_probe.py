"""GPU box: exact-fp32 3x3 / stride-1 convolutions of the benchmark (forward, mode 0, and data gradient, mode 1) in the pixel-patch
form (round 4: 128 x 64 tiles, dbn_set_patch_conv(1)) against the gather loop with the library's tile choice (dbn_set_patch_conv(2):
the 16-bit modes keep their patch kernels, exact fp32 gathers) and with an explicit 128 x 64 gather tile.  HIP-event timed, alone."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from gpu_util import L, rnd, DEV, igemm, pack
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(12):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[4]


STAGGERS = [int(v) for v in os.environ.get('STAGGERS', '0').split(',')]
shapes = ((16, 160, 64, 64, 'layer1 / FPN smooth p2'), (16, 80, 128, 128, 'layer2'), (16, 160, 256, 64, 'head 256->64'), (16, 80, 64, 64, 'FPN smooth p3'))
for (N, H, Ci, Co, what) in shapes:
    w = rnd(Co, Ci, 3, 3, seed=1, scale=0.05)
    flops = 2.0 * N * H * H * Ci * Co * 9
    for stg in STAGGERS:
        L().dbn_set_stagger(stg)
        for mode in (0, 1):
            cin, cout = (Ci, Co) if mode == 0 else (Co, Ci)
            x = torch.randn(N, H, H, cin, device=DEV)
            y = torch.empty(N, H, H, cout, device=DEV)
            wp = pack(w, mode)
            res = {}
            try:
                L().dbn_set_patch_conv(3)
                cfgp = L().dbn_igemm_kernel_config(0, 0, mode, N, H, H, cin, H, H, cout, 3, 3, 1, 1, 0, 1)
                res['patch'] = timed(lambda: igemm(x, wp, None, y, 3, 1, 1, mode))
                yp = y.clone()
                L().dbn_set_patch_conv(2)
                cfgg = L().dbn_igemm_kernel_config(0, 0, mode, N, H, H, cin, H, H, cout, 3, 3, 1, 1, 0, 1)
                res['gather(auto)'] = timed(lambda: igemm(x, wp, None, y, 3, 1, 1, mode))
                same = torch.equal(y, yp)
                res['gather 128x64'] = timed(lambda: igemm(x, wp, None, y, 3, 1, 1, mode, tile=3))
            finally:
                L().dbn_set_patch_conv(1)
            print('stagger %4d %-24s %3d->%3d @%3d mode %d: %s   [patch cfg %d, gather cfg %d, bit-identical %s]' % (
                stg, what, cin, cout, H, mode,
                '  '.join('%s %6.1f us %5.1f TF/s (%.3f)' % (k, v * 1e3, flops / v / 1e9, flops / v / 1e9 / 157.3) for k, v in res.items()),
                cfgp, cfgg, same))
L().dbn_set_stagger(0)
