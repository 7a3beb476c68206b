#!/usr/bin/env python3
"""GPU: WHICH output of dbn_head_tail_bwd_t changes between launches on fixed inputs beside a second process training on the same GPU
(tools/cotenancy.sh reproduces the effect: hundreds of distinct results in 600 launches; alone: one).  Starts the companion itself (a fresh
child process, before this process touches the GPU), launches the kernel `reps` times, compares every output with the first launch's
element by element.  usage: cotenancy_diff.py [f32|bf16] [reps] [companion math | none]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
at = 1 if (len(sys.argv) > 1 and sys.argv[1] == 'bf16') else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
comp = sys.argv[3] if len(sys.argv) > 3 else 'bf16'
child = None
if comp != 'none':
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, 'tools', 'cfg_timing.py'), 'resnet18', '16', '640', comp, '100000'],
                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    time.sleep(25)  # (import torch + model set-up + the first steps)
import torch
from db_text_minimal_amd import _lib
L = _lib.lib()
dev = 'cuda'
N, Hq, Wq = 16, 320, 320
g = torch.Generator(device=dev).manual_seed(1)
dt = torch.bfloat16 if at else torch.float32
xb = torch.randn(N, Hq, Wq, 64, device=dev, generator=g).to(dt)
xt = torch.randn(N, Hq, Wq, 64, device=dev, generator=g).to(dt)
wb, wt = torch.randn(256, device=dev, generator=g) * 0.2, torch.randn(256, device=dev, generator=g) * 0.2
preds = torch.rand(N, 3, 2 * Hq, 2 * Wq, device=dev, generator=g) * 0.98 + 0.01
dpreds = torch.randn(N, 3, 2 * Hq, 2 * Wq, device=dev, generator=g) * 1e-3
scb, shb, sct, sht = (torch.rand(64, device=dev, generator=g) + 0.5 for _ in range(4))
mub, rsb, mut, rst = (torch.rand(64, device=dev, generator=g) + 0.3 for _ in range(4))
ws = torch.empty(L.dbn_head_tail_bwd_ws_floats(), device=dev)
st = torch.cuda.current_stream().cuda_stream


def run():
    outs = dict(sums=torch.full((256, ), float('nan'), device=dev), dxb=torch.full((N, Hq, Wq, 64), float('nan'), device=dev, dtype=dt),
                dxt=torch.full((N, Hq, Wq, 64), float('nan'), device=dev, dtype=dt), dwb=torch.full((256, ), float('nan'), device=dev),
                dbb=torch.full((1, ), float('nan'), device=dev), dwt=torch.full((256, ), float('nan'), device=dev), dbt=torch.full((1, ), float('nan'), device=dev))
    ws.fill_(float('nan'))
    ws[770 * 2047:].zero_()  # (the A/B build's debug counters: DBN_HT_CHECK)
    _lib.check(L.dbn_head_tail_bwd_t(at, xb.data_ptr(), xt.data_ptr(), wb.data_ptr(), wt.data_ptr(), preds.data_ptr(), dpreds.data_ptr(),
                                     scb.data_ptr(), shb.data_ptr(), sct.data_ptr(), sht.data_ptr(), mub.data_ptr(), rsb.data_ptr(), mut.data_ptr(),
                                     rst.data_ptr(), outs['sums'].data_ptr(), outs['dxb'].data_ptr(), outs['dxt'].data_ptr(), outs['dwb'].data_ptr(),
                                     outs['dbb'].data_ptr(), outs['dwt'].data_ptr(), outs['dbt'].data_ptr(), N, Hq, Wq, 3, 50.0, 1.0, ws.data_ptr(), st), 'head_tail_bwd')
    torch.cuda.synchronize()
    outs['ws'] = ws[:770 * 2047].clone()
    outs['dbg'] = ws[770 * 2047:770 * 2047 + 8].clone()
    return outs


ref = run()
stats = {k: [0, 0, 0.0] for k in ref}  # launches that differ, elements that differ (sum), max |diff|
first_bad = None
dbg_tot = ref['dbg'].clone()
for r in range(reps):
    o = run()
    dbg_tot += o['dbg']
    for k in ref:
        a, b = ref[k].float(), o[k].float()
        ne = (a != b) & ~(torch.isnan(a) & torch.isnan(b))
        n = int(ne.sum())
        if n:
            stats[k][0] += 1
            stats[k][1] += n
            stats[k][2] = max(stats[k][2], float((a - b)[ne].abs().max()))
            if first_bad is None and k in ('dxb', 'ws'):
                idx = ne.reshape(-1).nonzero()[:8, 0].tolist()
                first_bad = (k, r, idx, [float(a.reshape(-1)[i]) for i in idx], [float(b.reshape(-1)[i]) for i in idx])
print('storage %s, %d launches after the reference one, companion: %s' % ('bf16' if at else 'f32', reps, comp))
for k, (nl, ne, mx) in stats.items():
    print('  %-5s launches that differ %4d   elements that differ (all launches) %10d   max |diff| %.3e   (tensor max %.3e)' % (k, nl, ne, mx, float(ref[k].float().nan_to_num().abs().max())))
print('  in-kernel register check (DBN_HT_CHECK builds; lanes over all launches): wbq %d, wtq %d, scale/shift %d, low-16-bits-only %d' % tuple(int(dbg_tot[i]) for i in (0, 1, 2, 4)))
print('  first difference in dxb / ws:', first_bad)
if first_bad is not None and first_bad[0] == 'dxb' and at == 0:
    # what WAS computed there?  dxb[px][4q + e] = sum_ab wb[(4q + e) * 4 + ab] * dl_b[ab]; rebuild dl_b of that pixel on the host and look for the
    # weight row whose product gives the observed value
    k, r, idx, good, badv = first_bad
    for i, gv, bv in list(zip(idx, good, badv))[:4]:
        px, c = i // 64, i % 64
        n_, rem = px // (Hq * Wq), px % (Hq * Wq)
        hq_, wq_ = rem // Wq, rem % Wq
        P = preds[n_, 0, 2 * hq_:2 * hq_ + 2, 2 * wq_:2 * wq_ + 2].reshape(-1).double()
        B = preds[n_, 2, 2 * hq_:2 * hq_ + 2, 2 * wq_:2 * wq_ + 2].reshape(-1).double()
        dP = dpreds[n_, 0, 2 * hq_:2 * hq_ + 2, 2 * wq_:2 * wq_ + 2].reshape(-1).double()
        dB = dpreds[n_, 2, 2 * hq_:2 * hq_ + 2, 2 * wq_:2 * wq_ + 2].reshape(-1).double()
        dl = (dP + dB * 50.0 * B * (1 - B)) * P * (1 - P)
        rows = (wb.double().view(64, 4) * dl.view(1, 4)).sum(1).cpu()
        near = lambda v: int((rows - v).abs().argmin())
        print('    element %d = pixel %d (n %d, hq %d, wq %d) channel %d: reference %.6e (weight row %d gives %.6e), observed %.6e (closest weight row %d: %.6e)'
              % (i, px, n_, hq_, wq_, c, gv, c, float(rows[c]), bv, near(bv), float(rows[near(bv)])))
        import itertools
        wrow = wb.double().view(64, 4)[c].cpu()
        dlc = dl.cpu()
        best = min(((abs(float((wrow * dlc[list(pm)]).sum()) - bv), pm) for pm in itertools.product(range(4), repeat=4)), key=lambda t: t[0])
        print('      same weight row with dl_b lanes %s instead of (0, 1, 2, 3): off by %.2e;  dl_b = %s' % (best[1], best[0], [float(v) for v in dlc]))
        # the same row with the dl_b of a NEIGHBOURING pixel (x +- 1, y +- 1)?
        for dyq, dxq in ((0, 1), (0, -1), (1, 0), (-1, 0)):
            h2, w2 = hq_ + dyq, wq_ + dxq
            if 0 <= h2 < Hq and 0 <= w2 < Wq:
                P2 = preds[n_, 0, 2 * h2:2 * h2 + 2, 2 * w2:2 * w2 + 2].reshape(-1).double(); B2 = preds[n_, 2, 2 * h2:2 * h2 + 2, 2 * w2:2 * w2 + 2].reshape(-1).double()
                dP2 = dpreds[n_, 0, 2 * h2:2 * h2 + 2, 2 * w2:2 * w2 + 2].reshape(-1).double(); dB2 = dpreds[n_, 2, 2 * h2:2 * h2 + 2, 2 * w2:2 * w2 + 2].reshape(-1).double()
                dl2 = ((dP2 + dB2 * 50.0 * B2 * (1 - B2)) * P2 * (1 - P2)).cpu()
                print('      with the dl_b of pixel (hq %+d, wq %+d): %.6e' % (dyq, dxq, float((wrow * dl2).sum())))

if child is not None:
    child.kill()
    child.wait()
