"""Debug helper (GPU box): activation-gradient error of the HIP path vs the CPU oracle (autograd)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
import torch
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
from oracle import dbnet_oracle as O

n, size, seed = int(sys.argv[1]), int(sys.argv[2]), 3
img, gts = O.synthetic_batch(n, size, seed=seed)
sd = O.new_state(seed)
m = DBTextModel(); m.load_state_dict(sd); m = m.cuda().train()
eng = m.engine
preds = eng.forward(img.cuda(), train=True)
tr = DBTrainer(m, DBLoss(), FusedAdam(m))
losses, dpreds = tr._loss(preds, gts.cuda())
eng.backward(dpreds)
torch.cuda.synchronize()
taps = {}
work = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and 'running' not in k else v.clone()) for k, v in sd.items()}
preds_o = O.forward(work, img, training=True, taps=taps)
preds_o.retain_grad()
for v in taps.values():
    v.retain_grad()
O.db_loss(preds_o, gts)[4].backward()
def cmp(tag, mine, ref):
    err = (mine - ref).abs()
    print('%-14s max|ref| %.3e max err %.3e rel %.3e  L2rel %.3e' % (tag, float(ref.abs().max()), float(err.max()), float(err.max() / ref.abs().max()), float(err.norm() / ref.norm())))
cmp('dpreds', dpreds.cpu(), preds_o.grad)
nhwc = lambda t: t.permute(0, 3, 1, 2).cpu()
for br in ('binarize', 'thresh'):
    cmp(br + ' dz1', nhwc(eng.bufs[br + '/dz1']), taps[br + '/z1'].grad)
    cmp(br + ' dy1', nhwc(eng.bufs[br + '/dy1']), taps[br + '/y1'].grad)
    cmp(br + ' dz0', nhwc(eng.bufs[br + '/dz0']), taps[br + '/z0'].grad)
    cmp(br + ' dy0', nhwc(eng.bufs[br + '/dy0']), taps[br + '/y0'].grad)
cmp('dfpn', nhwc(eng.bufs['fpn/dz']), taps['fpn'].grad)
for k, b in (('p2', 'smooth_p2/dz'), ('p3', 'smooth_p3/dz'), ('p4', 'smooth_p4/dz'), ('p5', 'reduce_conv_c5/dz'),
             ('c5', 'dbackbone.layer4.1/out'), ('c4', 'dbackbone.layer3.1/out'), ('c3', 'dbackbone.layer2.1/out'),
             ('c2', 'dbackbone.layer1.1/out'), ('pool', 'stem/dpool')):
    cmp('d' + k, nhwc(eng.bufs[b]), taps[k].grad)
for br in ('binarize', 'thresh'):
    mine, ref = nhwc(eng.bufs[br + '/dy1']), taps[br + '/y1'].grad
    e = (mine - ref).abs().amax((0, 2, 3)); r = ref.abs().amax((0, 2, 3))
    mu = eng.bufs['segmentation_head.%s.4/mean' % br].cpu(); rs = eng.bufs['segmentation_head.%s.4/rstd' % br].cpu()
    y1 = taps[br + '/y1']
    mu_o = y1.mean((0, 2, 3)); rs_o = 1 / torch.sqrt(y1.var((0, 2, 3), unbiased=False) + 1e-5)
    g = nhwc(eng.bufs[br + '/dz1']); go = taps[br + '/z1'].grad
    zm = (taps[br + '/z1'] > 0).float()
    print(br, 'worst channels:')
    for c in torch.argsort(e / r, descending=True)[:5].tolist():
        print('  c%02d relerr %.2e  mean %.4e (o %.4e) rstd %.4e (o %.4e)  sum g %.4e (o %.4e)  frac>0 %.3f' % (
            c, float(e[c] / r[c]), float(mu[c]), float(mu_o[c]), float(rs[c]), float(rs_o[c]),
            float((g[:, c] * zm[:, c]).sum()), float((go[:, c] * zm[:, c]).sum()), float(zm[:, c].mean())))
br = 'binarize'
y1 = taps[br + '/y1'].detach().double(); go = (taps[br + '/z1'].grad * (taps[br + '/z1'] > 0)).double()
gam = sd['segmentation_head.binarize.4.weight'].double().view(1, -1, 1, 1)
mu = y1.mean((0, 2, 3), keepdim=True); var = y1.var((0, 2, 3), unbiased=False, keepdim=True); rs = 1 / torch.sqrt(var + 1e-5)
xh = (y1 - mu) * rs
c1 = go.mean((0, 2, 3), keepdim=True); c2 = (go * xh).mean((0, 2, 3), keepdim=True)
dy64 = gam * rs * (go - c1 - xh * c2)
mine, ref = nhwc(eng.bufs[br + '/dy1']).double(), taps[br + '/y1'].grad.double()
for c in (60, 40, 20):
    print('c%d: |dy64| max %.3e  oracle-vs-64 %.3e  mine-vs-64 %.3e   c1 %.4e c2 %.4e  max|g| %.3e' % (
        c, float(dy64[:, c].abs().max()), float((ref[:, c] - dy64[:, c]).abs().max()), float((mine[:, c] - dy64[:, c]).abs().max()),
        float(c1[0, c]), float(c2[0, c]), float(go[:, c].abs().max())))
dg = eng.grad_views['segmentation_head.binarize.4.weight'].cpu().double(); db = eng.grad_views['segmentation_head.binarize.4.bias'].cpu().double()
M = y1.numel() / 64
for c in (60, 40, 20):
    print('c%d dgamma mine %.6e ref %.6e | dbeta mine %.6e ref %.6e' % (c, float(dg[c]), float(c2[0, c] * M), float(db[c]), float(c1[0, c] * M)))
