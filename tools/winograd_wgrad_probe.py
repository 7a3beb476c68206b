"""GPU box: Winograd F(2x2,3x3) weight gradient (winograd_wgrad_f32_kernel + its slab reduction) against the direct exact-fp32 kernels
on the benchmark's 3x3 / stride-1 layer shapes; alone, HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from db_text_minimal_amd import _lib
from gpu_util import L, DEV, stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn):
    """ms per launch, back to back with settled clocks (round 6: a launch timed alone after a synchronize, 5 warm-up launches before, read
    10-25 % slow — the clocks ramp for tens of milliseconds; tools/wres_probe.py)."""
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    for _ in range(max(20, int(60.0 / max(e0.elapsed_time(e1) / 5, 1e-3)))):  # >= 60 ms of launches
        fn()
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ts.sort()
    return ts[2]


for (N, H, Ci, Co, what) in ((16, 160, 64, 64, 'layer1 / smooth_p2'), (16, 160, 256, 64, 'head 256->64'), (16, 80, 128, 128, 'layer2'),
                             (16, 80, 64, 64, 'smooth_p3'), (16, 40, 256, 256, 'layer3'), (16, 20, 512, 512, 'layer4')):
    x = torch.randn(N, H, H, Ci, device=DEV)
    dy = torch.randn(N, H, H, Co, device=DEV)
    g = torch.empty(Co, Ci, 3, 3, device=DEV)
    slab = torch.empty(L().dbn_winograd_wgrad_slab_floats(N, H, H, Co, Ci), device=DEV)
    a = (dy.data_ptr(), x.data_ptr(), None, None, slab.data_ptr(), g.data_ptr(), N, H, H, Co, Ci, Ci, 1.0, stream())
    t1 = timed(lambda: _lib.check(L().dbn_winograd_wgrad_f32(1, *a), 'w1'))
    t2 = timed(lambda: _lib.check(L().dbn_winograd_wgrad_f32(2, *a), 'w2'))
    gw = g.clone()
    slabd = torch.empty(L().dbn_wgrad_slab_floats_hw(N, H, H, Co, H, H, Ci, 3, 3, 4), device=DEV)
    ad = (0, 0, dy.data_ptr(), x.data_ptr(), slabd.data_ptr(), g.data_ptr(), N, H, H, Co, H, H, Ci, Ci, 3, 3, 1, 1, 1.0, stream())
    d1 = timed(lambda: _lib.check(L().dbn_wgrad_phase_t(1, *ad), 'd1'))
    d2 = timed(lambda: _lib.check(L().dbn_wgrad_phase_t(2, *ad), 'd2'))
    flops = 2.0 * N * H * H * Ci * Co * 9
    err = float((gw - g).abs().max() / g.abs().max())
    print('%-20s %3d->%3d @%3d: winograd %7.1f + %5.1f us (MFMA pipe %.3f, slab %5.1f MB)   direct %7.1f + %5.1f us (%.3f)   speed-up %.2fx   max rel diff %.1e' % (
        what, Ci, Co, H, t1 * 1e3, t2 * 1e3, flops * 4 / 9 / t1 / 1e9 / 157.3, slab.numel() * 4 / 1e6, d1 * 1e3, d2 * 1e3, flops / d1 / 1e9 / 157.3, (d1 + d2) / (t1 + t2), err))
