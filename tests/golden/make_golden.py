#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference; nothing in the test
suite or on the GPU box does):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's DBTextModel / DBLoss (models.py, losses.py) with
`model_zoo.load_url` stubbed to `{}` (no network; models.py:17 hard-codes
pretrained=True), fills the model with `oracle.dbnet_oracle.procedural_fill`
(a build-owned per-key seeded fill, so weights need not be committed), drives
it with a 10-line restatement of train.py:160-172 and stores inputs'
seeds + expected outputs as small .npz files.  The fixtures are data only.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference/src')
sys.dont_write_bytecode = True

import torch.utils.model_zoo as mz  # noqa: E402

mz.load_url = lambda *a, **k: {}

from models import DBTextModel  # noqa: E402  (reference)
from losses import DBLoss  # noqa: E402  (reference)
from oracle import dbnet_oracle as O  # noqa: E402

torch.set_num_threads(8)
SAMPLE = 256


def sample_idx(numel, k=SAMPLE):
    if numel <= k:
        return np.arange(numel)
    return (np.arange(k, dtype=np.int64) * (numel // k)) + (numel // (2 * k))


def summarize(prefix, t, out, full_below=8192, k=SAMPLE):
    a = t.detach().double().reshape(-1).numpy()
    out[prefix + '/stats'] = np.array([a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum()), a.min(), a.max()])
    if a.size <= full_below:
        out[prefix + '/full'] = t.detach().float().numpy()
    else:
        out[prefix + '/sample'] = a[sample_idx(a.size, k)].astype(np.float32)


def make_ref(seed, arch='resnet18'):
    """The reference's DBTextModel.  models.py:8 registers only resnet18 although resnet.py:285-306 defines the
    Bottleneck nets (SURVEY A4'): for 'resnet50' the registry entry is swapped for the duration of the constructor, so
    the reference's own __init__/forward assemble resnet50 + FPN([256,512,1024,2048]) + DBHead.  (The deformable
    variants import torchvision inside the block constructor and cannot be built here.)"""
    import models as ref_models
    from modules.resnet import resnet50
    saved = dict(ref_models.backbone_dict['resnet18'])
    try:
        if arch == 'resnet50':
            ref_models.backbone_dict['resnet18'] = {'models': resnet50, 'out': [256, 512, 1024, 2048]}
        else:
            assert arch == 'resnet18'
        m = DBTextModel()
    finally:
        ref_models.backbone_dict['resnet18'] = saved
    O.procedural_fill(m.state_dict(), seed)
    return m


def case_train(name, n, size, seed, img_scale=1.0, full_maps=True, steps=1, arch='resnet18', map_sample=SAMPLE):
    print('==', name)
    torch.manual_seed(0)
    m = make_ref(seed, arch).train()
    crit = DBLoss(alpha=1.0, beta=10.0, reduction='mean', negative_ratio=3)
    opt = torch.optim.Adam(m.parameters(), lr=0.005, weight_decay=0, amsgrad=False)
    img, gts = O.synthetic_batch(n, size, seed=seed + 100, img_scale=img_scale)
    out = {'meta': np.array([n, size, seed, steps]), 'img_scale': np.array(img_scale)}
    step_losses = []
    for it in range(steps):
        preds = m(img)
        assert preds.size(1) == 3
        losses = crit(preds, gts)
        opt.zero_grad()
        losses[4].backward()
        if it == 0:
            if full_maps:
                out['preds'] = preds.detach().numpy()
            else:
                for c, nm in enumerate('PTB'):
                    summarize('preds_' + nm, preds[:, c], out, full_below=0, k=map_sample)
            for k, p in m.named_parameters():
                if p.grad is not None:
                    summarize('grad/' + k, p.grad, out)
            out['dead_grad_none'] = np.array(
                [int(p.grad is None) for k, p in m.named_parameters() if k.startswith(('backbone.fc', 'backbone.smooth'))])
        opt.step()
        step_losses.append([float(v) for v in losses])
        print('  step', it, step_losses[-1])
    out['losses'] = np.array(step_losses)
    sd = m.state_dict()
    # post-step state: running stats always, params as summaries
    for k, v in sd.items():
        if 'running' in k:
            summarize('post/' + k, v, out, full_below=600)
        elif 'num_batches' not in k and not k.startswith(('backbone.fc', 'backbone.smooth')):
            summarize('post/' + k, v, out, full_below=0)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    return m


def case_fp64(name, n, size, seed, arch='resnet18', bn3_gain=None):
    """The reference evaluated in DOUBLE (model.double()) next to its own fp32 run AND its own run under
    torch.autocast('cpu', dtype=torch.bfloat16) (the forward in autocast, DBLoss on preds.float() — how the reference would
    be run in bf16), same weights and inputs.  Stored: the fp64 losses / maps (strided sample) / per-parameter gradients
    (norm + strided sample), and for the fp32 and the bf16-autocast run the losses, the same map samples and the per-parameter
    distance of the gradient from the fp64 one.  The GPU tests require the HIP paths to be no further from fp64 than a small
    multiple of what the reference's own arithmetic at that precision is (tests/test_model_gpu.py)."""
    print('==', name)
    img, gts = O.synthetic_batch(n, size, seed=seed + 100)
    grads = {}
    out = {'meta': np.array([n, size, seed, 1]), 'bn3_gain': np.array(1.0 if bn3_gain is None else bn3_gain)}
    for tag, dt in (('f32', torch.float32), ('f64', torch.float64), ('bf16ac', torch.float32)):
        m = make_ref(seed, arch)
        if bn3_gain is not None:  # the same conditioning the GPU tests apply to the 53-layer nets (tools/bf16_dcn_probe.py)
            with torch.no_grad():
                for k, p in m.named_parameters():
                    if k.endswith('bn3.weight'):
                        p.mul_(bn3_gain)
        m = m.to(dt).train()
        if tag == 'bf16ac':
            with torch.autocast('cpu', dtype=torch.bfloat16):
                preds = m(img)
            preds = preds.float()
        else:
            preds = m(img.to(dt))
        losses = DBLoss()(preds, gts.to(dt))
        losses[4].backward()
        grads[tag] = {k: p.grad.detach().double() for k, p in m.named_parameters() if p.grad is not None}
        out['losses_' + tag] = np.array([float(v) for v in losses])
        out['preds_' + tag + '/sample'] = preds.detach().double().reshape(-1).numpy()[sample_idx(preds.numel(), 16384)]
    for k, g64 in grads['f64'].items():
        a = g64.reshape(-1).numpy()
        out['g64/' + k + '/norm'] = np.array(np.sqrt((a * a).sum()))
        out['g64/' + k + '/sample'] = a[sample_idx(a.size)]
        out['ref32_dist/' + k] = np.array(float((grads['f32'][k] - g64).norm()))
        out['refbf16_dist/' + k] = np.array(float((grads['bf16ac'][k] - g64).norm()))
        out['refbf16_cos/' + k] = np.array(float((grads['bf16ac'][k] * g64).sum() / (grads['bf16ac'][k].norm() * g64.norm() + 1e-300)))
    for tag in ('ref32_dist', 'refbf16_dist'):
        worst = max(float(out[tag + '/' + k] / (out['g64/' + k + '/norm'] + 1e-300)) for k in grads['f64'])
        tot = np.sqrt(sum(float(out[tag + '/' + k])**2 for k in grads['f64']) / sum(float(out['g64/' + k + '/norm'])**2 for k in grads['f64']))
        print('  %s: worst per-tensor |g - g64| / |g64| = %.3e, whole model %.3e' % (tag, worst, tot))
    for tag in ('f32', 'bf16ac'):
        d = np.abs(out['preds_' + tag + '/sample'] - out['preds_f64/sample'])
        print('  maps %s vs f64: max %.3e mean %.3e; losses %s' % (tag, d.max(), d.mean(), out['losses_' + tag] - out['losses_f64']))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


def case_odd(name, n, h, w, seed):
    """A size that is NOT a multiple of 32: the FPN's size-based nearest upsampling (segmentation_body.py:64-76) and the final
    F.interpolate(bilinear, align_corners=True) of models.py:43-46 are real resamples (the inference CLIs resize without padding,
    utils.py:160-175).  The reference's eval forward (+ DBLoss single value) and one train step: full maps, losses, gradients,
    running statistics."""
    print('==', name)
    img, gts = O.synthetic_batch(n, (h, w), seed=seed + 100)
    out = {'meta': np.array([n, h, w, seed])}
    m = make_ref(seed).eval()
    with torch.no_grad():
        pe = m(img)
        assert pe.shape == (n, 2, h, w)
        out['eval_preds'] = pe.numpy()
        out['eval_loss'] = np.array(float(DBLoss()(pe, gts)))
    m = make_ref(seed).train()
    preds = m(img)
    assert preds.shape == (n, 3, h, w)
    losses = DBLoss()(preds, gts)
    losses[4].backward()
    out['preds'] = preds.detach().numpy()
    out['losses'] = np.array([float(v) for v in losses])
    for k, p in m.named_parameters():
        if p.grad is not None:
            summarize('grad/' + k, p.grad, out)
    for k, v in m.state_dict().items():
        if 'running' in k:
            summarize('post/' + k, v, out, full_below=600)
    print('  losses', out['losses'], 'eval loss', out['eval_loss'])
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


def case_eval(name, n, size, seed):
    print('==', name)
    m = make_ref(seed).eval()
    img, gts = O.synthetic_batch(n, size, seed=seed + 100)
    with torch.no_grad():
        preds = m(img)
        assert preds.size(1) == 2
        val = DBLoss()(preds, gts)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), meta=np.array([n, size, seed, 0]), preds=preds.numpy(),
                        loss=np.array(float(val)))


def case_loss_kats():
    print('== loss_kats')
    out = {}
    crit = DBLoss()
    g = torch.Generator().manual_seed(7)

    def run(tag, preds, gts, crit_=crit):
        preds = preds.clone().requires_grad_(True)
        res = crit_(preds, gts)
        res5 = res if isinstance(res, tuple) else (res, )
        tot = res5[-1]
        (dp, ) = torch.autograd.grad(tot, preds)
        out[tag + '/preds'] = preds.detach().numpy()
        out[tag + '/gts'] = gts.numpy()
        out[tag + '/losses'] = np.array([float(v) for v in res5])
        out[tag + '/dpreds'] = dp.numpy()
        print('  ', tag, out[tag + '/losses'])

    n, s = 2, 64
    P = torch.rand(n, 1, s, s, generator=g) * 0.98 + 0.01
    T = torch.rand(n, 1, s, s, generator=g) * 0.98 + 0.01
    B = torch.reciprocal(1 + torch.exp(-50 * (P - T)))
    preds3 = torch.cat([P, T, B], 1)
    _, gts = O.synthetic_batch(n, s, seed=11)
    run('default', preds3, gts)
    run('eval2ch', preds3[:, :2].contiguous(), gts)
    g0 = gts.clone()
    g0[0].zero_()  # no positives -> prob_loss 0.0 (topk k=0)
    run('no_positive', preds3, g0)
    g1 = gts.clone()
    g1[1].zero_()  # everything masked out
    g1[3].zero_()
    run('all_masked', preds3, g1)
    # few positives so that n_neg = 3*n_pos side of the min() is taken, and the other side
    g2 = gts.clone()
    g2[0] = (torch.rand(n, s, s, generator=g) > 0.5).float()
    run('neg_limited', preds3, g2)
    # saturated probabilities: log clamp at -100
    Ps = (torch.rand(n, 1, s, s, generator=g) > 0.5).float()
    run('saturated', torch.cat([Ps, T, torch.reciprocal(1 + torch.exp(-50 * (Ps - T)))], 1), gts)
    run('alpha_beta', preds3, gts, DBLoss(alpha=5.0, beta=2.0, negative_ratio=1))
    # contrast: reduction='none' (true per-pixel OHEM) — recorded so a future top-k kernel can be pinned
    run('reduction_none', preds3, gts, DBLoss(reduction='none'))
    # losses.py:30 hands any torch reduction string to F.binary_cross_entropy: 'sum' = the scalar BCE summed, not averaged
    run('reduction_sum', preds3, gts, DBLoss(reduction='sum'))
    # NON-BINARY maps (round 6): losses.py:33-39 takes topk over loss * negative literally, which differs from the binary-map
    # closed form bce * n_neg.  (i) a fractional supervision_mask, (ii) fractional prob_gt AND mask, (iii) few positives so that
    # n_neg = int(ratio * n_pos) < sum(negative) and the top-k really selects, (iv) the same under reduction='sum'.
    # (drawn after every older case: the older cases' random streams are unchanged)
    gf = gts.clone()
    gf[1] = torch.rand(n, s, s, generator=g)
    run('fractional_mask', preds3, gf)
    gf2 = gts.clone()
    gf2[0] = torch.rand(n, s, s, generator=g) ** 3
    gf2[1] = torch.rand(n, s, s, generator=g)
    run('fractional_gt_and_mask', preds3, gf2)
    gf3 = gts.clone()
    gf3[0] = (torch.rand(n, s, s, generator=g) > 0.97).float()
    gf3[1] = torch.rand(n, s, s, generator=g)
    run('fractional_mask_topk_selects', preds3, gf3, DBLoss(negative_ratio=2))
    run('fractional_mask_sum', preds3, gf3, DBLoss(reduction='sum', negative_ratio=2))
    np.savez_compressed(os.path.join(HERE, 'loss_kats.npz'), **out)


def case_dp(name, size, seed):
    """gradient after all-reduce == mean of per-shard grads."""
    print('==', name)
    img, gts = O.synthetic_batch(2, size, seed=seed + 100)
    acc = None
    out = {'meta': np.array([2, size, seed, 1])}
    for r in range(2):
        m = make_ref(seed).train()
        crit = DBLoss()
        preds = m(img[r:r + 1])
        tot = crit(preds, gts[:, r:r + 1])[4]
        tot.backward()
        out['loss_rank%d' % r] = np.array(float(tot))
        gr = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
        acc = gr if acc is None else {k: acc[k] + gr[k] for k in acc}
    for k, v in acc.items():
        summarize('grad/' + k, v / 2, out)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


def case_pixel_metrics():
    """cal_text_score / RunningScore of the reference (text_metrics.py); its imports `iou` (shapely) and
    `utils` (cv2, ...) are not installed here and are unrelated to the pixel metric -> stubbed."""
    print('== pixel_metrics')
    import types
    for name, attrs in (('iou', {'DetectionIoUEvaluator': object}), ('utils', {'to_list_tuples_coords': None})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    from text_metrics import RunningScore, cal_text_score  # reference
    g = torch.Generator().manual_seed(21)
    out = {}
    rs = RunningScore(2)
    for step in range(2):
        P = torch.rand(2, 48, 40, generator=g)
        G = (torch.rand(2, 48, 40, generator=g) > 0.7).float()
        M = (torch.rand(2, 48, 40, generator=g) > 0.1).float()
        score = cal_text_score(P, G, M, rs, thresh=0.25)
        out['step%d/P' % step], out['step%d/G' % step], out['step%d/M' % step] = P.numpy(), G.numpy(), M.numpy()
        out['step%d/hist' % step] = rs.confusion_matrix.copy()
        out['step%d/scores' % step] = np.array([score[k] for k in ('Overall Acc', 'Mean Acc', 'FreqW Acc', 'Mean IoU')])
        hist_o = O.pixel_confusion(P, G, M, 0.25)
        print('  ', rs.confusion_matrix.tolist(), score)
    np.savez_compressed(os.path.join(HERE, 'pixel_metrics.npz'), **out)


def check_oracle(arch='resnet18'):
    """Pin the oracle against the imported reference right here."""
    print('== oracle vs reference', arch)
    m = make_ref(3, arch).train()
    sd = O.new_state(3, arch)
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    assert list(m.state_dict().keys()) == list(sd.keys())
    img, gts = O.synthetic_batch(2, 96, seed=5)
    preds = m(img)
    losses = DBLoss()(preds, gts)
    losses[4].backward()
    p2, l2, g2 = O.loss_and_grads(sd, img, gts)
    print('  preds maxdiff', float((preds - p2).abs().max()))
    print('  losses', [float(v) for v in losses], l2)
    worst = 0.0
    for k, p in m.named_parameters():
        if p.grad is None:
            assert k not in g2 or g2[k] is None
            continue
        d = float((p.grad - g2[k]).abs().max() / (p.grad.abs().max() + 1e-12))
        worst = max(worst, d)
    print('  worst rel grad diff', worst)
    for k, v in m.state_dict().items():
        if 'running' in k:
            assert torch.allclose(v, sd[k], atol=1e-6), k
    assert float((preds - p2).abs().max()) < 1e-5 and worst < 1e-4


if __name__ == '__main__':
    if '--only-cfg2' in sys.argv:  # BASELINE configs[1]: the benchmarked workload itself, one train step of the reference
        case_train('cfg2_16x640', 16, 640, seed=16, full_maps=False, steps=1, map_sample=4096)
        sys.exit(0)
    if '--only-kats' in sys.argv:
        case_loss_kats()
        sys.exit(0)
    if '--only-odd' in sys.argv:
        case_odd('odd_1x96x70', 1, 96, 70, seed=8)
        sys.exit(0)
    if '--only-fp64' in sys.argv:
        case_fp64('fp64_2x128', 2, 128, seed=2)
        case_fp64('fp64_r50_2x96', 2, 96, seed=12, arch='resnet50')
        case_fp64('fp64_r50_2x96_bn3x02', 2, 96, seed=12, arch='resnet50', bn3_gain=0.2)
        sys.exit(0)
    if '--only-r50' in sys.argv:
        check_oracle('resnet50')
        case_train('r50_train_1x128', 1, 128, seed=11, steps=2, arch='resnet50')
        case_train('r50_train_2x96', 2, 96, seed=12, steps=1, arch='resnet50')
        sys.exit(0)
    check_oracle()
    case_pixel_metrics()
    if '--only-metrics' in sys.argv:
        sys.exit(0)
    case_loss_kats()
    case_train('train_1x64', 1, 64, seed=1, steps=3)
    case_train('train_2x128', 2, 128, seed=2, steps=3)
    case_train('train_2x96_scaled', 2, 96, seed=4, img_scale=60.0, steps=1)
    case_eval('eval_2x128', 2, 128, seed=2)
    case_odd('odd_1x96x70', 1, 96, 70, seed=8)
    case_dp('dp_2x1x128', 128, seed=6)
    check_oracle('resnet50')
    case_train('r50_train_1x128', 1, 128, seed=11, steps=2, arch='resnet50')
    case_train('r50_train_2x96', 2, 96, seed=12, steps=1, arch='resnet50')
    case_fp64('fp64_2x128', 2, 128, seed=2)
    case_fp64('fp64_r50_2x96', 2, 96, seed=12, arch='resnet50')
    case_fp64('fp64_r50_2x96_bn3x02', 2, 96, seed=12, arch='resnet50', bn3_gain=0.2)
    if '--no-640' not in sys.argv:
        case_train('cfg1_2x640', 2, 640, seed=0, full_maps=False, steps=3)
        case_train('cfg2_16x640', 16, 640, seed=16, full_maps=False, steps=1, map_sample=4096)
    print('done')
