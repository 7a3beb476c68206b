"""Sustained shader clock under pure fp32-MFMA load vs under an HBM-bound kernel vs idle: dbn_clock_probe (s_memtime /
s_memrealtime, one wave on its own stream) sampled while the main stream runs (a) the FPN pyramid conv back to back,
(b) bn_apply over a 420 MB tensor back to back, (c) nothing.  Usage (GPU box): python tools/clock_under_load.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam, _lib  # noqa: E402

dev = torch.device('cuda', 0)
L = _lib.lib()
model = DBTextModel().to(dev).train()
tr = DBTrainer(model, DBLoss(), FusedAdam(model))
g = torch.Generator(device=dev).manual_seed(1)
img = torch.randn(16, 3, 640, 640, device=dev, generator=g)
u = torch.rand(4, 16, 640, 640, device=dev, generator=g)
gts = torch.stack([(u[0] > 0.9).float(), (u[1] > 0.05).float(), 0.3 + 0.4 * u[2], (u[3] > 0.8).float()])
for _ in range(3):
    tr.step(img, gts)
torch.cuda.synchronize()
eng = model.engine
side = torch.cuda.Stream(device=dev)
khz = L.dbn_wall_clock_khz()


def probe(buf, us=200):
    n = buf.shape[0]
    for i in range(n):
        L.dbn_clock_probe(buf[i].data_ptr(), us, side.cuda_stream)
    return buf


def mhz(buf):
    v = buf.cpu().double()
    m = sorted((v[:, 0] / v[:, 1] * khz / 1000).tolist())
    return 'median %.0f  min %.0f  max %.0f MHz (%d samples)' % (m[len(m) // 2], m[0], m[-1], len(m))


def load_mfma(reps):
    fpn = model.segmentation_body
    zs = [eng.bufs[n + '/z'] for n in ('smooth_p2', 'smooth_p3', 'smooth_p4', 'reduce_conv_c5')]
    for _ in range(reps):
        eng._fpn_conv_forward('segmentation_body.conv.0', fpn.conv[0], zs, 'fpn/y', 'segmentation_body.conv.1', fpn.conv[1], True)


def load_hbm(reps):
    y = eng.bufs['fpn/y']
    sc, sh = eng.bufs['segmentation_body.conv.1/scale'], eng.bufs['segmentation_body.conv.1/shift']
    for _ in range(reps):
        eng.bn_apply(y, sc, sh, 'fpn/z')


for name, fn, reps in (('fp32 MFMA (pyramid conv, 1.85 ms x 40)', load_mfma, 40), ('HBM-bound (bn_apply 840 MB x 300)', load_hbm, 300),
                       ('whole train step x 3', lambda r: [tr.step(img, gts) for _ in range(r)], 3)):
    b = torch.zeros(60 if 'MFMA' in name else 40, 2, dtype=torch.int64, device=dev)  # (before the load is queued: same stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(reps)
    e1.record()
    probe(b)
    torch.cuda.synchronize()
    print('%-45s %s   [load ran %.1f ms]' % (name, mhz(b), e0.elapsed_time(e1)))
b = torch.zeros(10, 2, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
probe(b)
torch.cuda.synchronize()
print('%-45s %s' % ('idle', mhz(b)))
