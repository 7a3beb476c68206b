"""GPU: the adjoint of the deformable sampling — round 3's fixed-point scatter (dbn_deform_col2im_t) against round 5's gather
(dbn_deform_col2im_gather_t), alone, HIP events, on configs[3]'s three deformable stages (8 x 800^2: 100^2 x 128, 50^2 x 256, 25^2 x 512) for
normally distributed offsets of growing size and for one outlier in an otherwise quiet map.
usage: python tools/dcn_probe.py [f32|bf16] [H] [sigma]     (H, sigma: only that stage / offset size — for a rocprofv3 kernel trace)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from db_text_minimal_amd import _lib

L = _lib.lib()
dev = torch.device('cuda')
dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == 'bf16' else torch.float32
at = 1 if dt == torch.bfloat16 else 0
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


only_h = int(sys.argv[2]) if len(sys.argv) > 2 else 0
only_s = sys.argv[3] if len(sys.argv) > 3 else ''
for (N, C, H, stride) in ((8, 128, 100, 1), (8, 256, 50, 1), (8, 512, 25, 1), (8, 128, 200, 2)):
    if only_h and H != only_h:
        continue
    W = H
    Ho = Wo = (H + 2 - 3) // stride + 1
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, W, C, generator=g).to(dev).to(dt)
    dcols = torch.randn(N, Ho, Wo, 9 * C, generator=g).to(dev).to(dt)
    dx = torch.empty(N, H, W, C, device=dev, dtype=dt)
    doff = torch.empty(N, Ho, Wo, 64, device=dev, dtype=dt)
    ws3 = torch.empty(L.dbn_deform_col2im_ws_bytes(N, H, W, C, Ho, Wo, 3, 3), device=dev, dtype=torch.uint8)
    wsg = torch.empty(L.dbn_deform_col2im_gather_ws_bytes(N, Ho, Wo), device=dev, dtype=torch.uint8)
    dims = (N, H, W, C, Ho, Wo, 3, 3, stride, 1, 64)
    for label, scale, outlier in (('0', 0.0, 0), ('0.5', 0.5, 0), ('1', 1.0, 0), ('2', 2.0, 0), ('4', 4.0, 0), ('8', 8.0, 0), ('0.5 + one offset of 30', 0.5, 30)):
        if only_s and label != only_s:
            continue
        off = torch.zeros(N, Ho, Wo, 64)
        off[..., :18] = torch.randn(N, Ho, Wo, 18, generator=g) * scale
        if outlier:
            off[0, Ho // 2, Wo // 2, 3] = float(outlier)
        off = off.to(dev).to(dt)
        t3 = timed(lambda: _lib.check(L.dbn_deform_col2im_t(at, dcols.data_ptr(), x.data_ptr(), off.data_ptr(), dx.data_ptr(), doff.data_ptr(), 0,
                                                           ws3.data_ptr(), *dims, st), 'scatter'))
        tg = timed(lambda: _lib.check(L.dbn_deform_col2im_gather_t(at, dcols.data_ptr(), x.data_ptr(), off.data_ptr(), dx.data_ptr(), doff.data_ptr(),
                                                                  0, wsg.data_ptr(), *dims, st), 'gather'))
        print('%d x %d^2 x %d stride %d %s, offsets ~ N(0, %s): scatter %7.1f us   gather %7.1f us   (max |offset| %.1f)' %
              (N, H, C, stride, str(dt).split('.')[1], label, t3, tg, float(off.float().abs().max())))
