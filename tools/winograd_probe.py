"""GPU box: Winograd F(2x2,3x3) forward convolution (winograd_f32_kernel) against the direct exact-fp32 kernels on the benchmark's
3x3 / stride-1 layer shapes, both with the fused BatchNorm statistics; alone, HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from db_text_minimal_amd import _lib
from gpu_util import L, rnd, DEV, pack, stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
if 'DBN_WINO_PERSISTENT' in os.environ:  # A/B: 0 = one workgroup per item (round 4's form)
    L().dbn_set_winograd_persistent(int(os.environ['DBN_WINO_PERSISTENT']))
if 'DBN_WINO_CBS' in os.environ:
    L().dbn_set_winograd_blocks_per_barrier(int(os.environ['DBN_WINO_CBS']))
if 'DBN_WINO_STAGGER' in os.environ:
    L().dbn_set_winograd_stagger(int(os.environ['DBN_WINO_STAGGER']))


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
                             (16, 80, 64, 64, 'smooth_p3'), (16, 160, 64, 256, '(dgrad-shaped) 64->256'), (16, 40, 256, 256, 'layer3'), (16, 20, 512, 512, 'layer4')):
    w = rnd(Co, Ci, 3, 3, seed=1, scale=0.05)
    x = torch.randn(N, H, H, Ci, device=DEV)
    y = torch.empty(N, H, H, Co, device=DEV)
    gamma, beta = torch.ones(Co, device=DEV), torch.zeros(Co, device=DEV)
    rm, rv = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
    sc, sh, mu, rs = (torch.empty(Co, device=DEV) for _ in range(4))
    up = torch.empty(L().dbn_winograd_panel_floats(Co, Ci), device=DEV)
    _lib.check(L().dbn_winograd_pack(w.to(DEV).data_ptr(), Co, Ci, Ci, 0, up.data_ptr(), stream()), 'pack')
    wsw = torch.empty(L().dbn_winograd_ws_floats(N, H, H, Co), device=DEV)
    wp = pack(w, 0)
    wsd = torch.empty(L().dbn_conv_bn_ws_floats(N, H, H, Co, 0, 1), device=DEV)
    bn = (gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr())
    tw = timed(lambda: _lib.check(L().dbn_winograd_conv_bn_f32(x.data_ptr(), up.data_ptr(), None, y.data_ptr(), N, H, H, Ci, Co, *bn, wsw.data_ptr(), stream()), 'w'))
    yw = y.clone()
    td = timed(lambda: _lib.check(L().dbn_conv_bn_f32(x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, H, Ci, H, H, Co, 3, 3, 1, 1, 0, 0, 0, 0, *bn, wsd.data_ptr(), stream()), 'd'))
    flops = 2.0 * N * H * H * Ci * Co * 9
    err = float((yw - y).abs().max() / y.abs().max())
    print('%-24s %3d->%3d @%3d: winograd %7.1f us (%5.1f effective TF/s, MFMA pipe %.3f)   direct %7.1f us (%5.1f TF/s, %.3f)   speed-up %.2fx   max rel diff %.1e' % (
        what, Ci, Co, H, tw * 1e3, flops / tw / 1e9, flops * 4 / 9 / tw / 1e9 / 157.3, td * 1e3, flops / td / 1e9, flops / td / 1e9 / 157.3, td / tw, err))
