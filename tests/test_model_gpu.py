"""-m gpu: DBTextModel / DBLoss / per-step loop on MI355X vs (a) golden vectors produced by the
reference itself and (b) the CPU oracle run side by side.  north_star tolerance on the three
output maps: 1e-3 abs / 1e-2 rel (fp32).

What can and cannot be compared (measured, see DESIGN.md "Parity"):
  * step-0 maps and losses: per element / tight.
  * step-0 gradients: two fp32 implementations flip the ReLU mask of the few activations that are
    within round-off of zero (~1e-6..1e-5 of all elements); each flip moves one element's gradient
    by 100 %, so gradients agree to ~0.5 % in L2 (cosine > 0.9999), not per element to 1e-5.
  * steps >= 1: Adam's first update is lr*sign(g), so round-off-level gradient differences move
    individual weights by 2*lr; the maps of later steps are chaotic per pixel (the CPU reference
    shows the same when its own gradients are perturbed by 1e-6), the five losses are not.
"""
import os

import numpy as np
import pytest
import torch

from gpu_util import DEV, report, report_robust
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
from oracle import dbnet_oracle as O

pytestmark = pytest.mark.gpu

MAP_ATOL, MAP_RTOL = 1e-3, 1e-2  # BASELINE.json north_star


def make_model(seed, arch='resnet18'):
    m = DBTextModel() if arch == 'resnet18' else DBTextModel(arch)
    m.load_state_dict(O.new_state(seed, arch))
    return m.to(DEV)


def sample_idx(numel, k=256):
    if numel <= k:
        return np.arange(numel)
    return (np.arange(k, dtype=np.int64) * (numel // k)) + (numel // (2 * k))


def check_grad_summary(z, prefix, t, l2_rtol=2e-2, sample_tol=5e-2, cos_min=0.999):
    """Gradient vs golden summary: L2 norm and a strided sample, at ReLU-flip-level tolerances."""
    a = t.detach().double().cpu().reshape(-1)
    st = z[prefix + '/stats']
    scale = max(float(st[4]), -float(st[3]), 1e-30)
    key = prefix + ('/full' if prefix + '/full' in z.files else '/sample')
    ref = torch.from_numpy(z[key]).double().reshape(-1)
    got = a if key.endswith('full') else a[torch.from_numpy(sample_idx(a.numel()))]
    err = float((got - ref).abs().max())
    cos = float((got @ ref) / (got.norm() * ref.norm())) if float(ref.norm()) > 0 and float(got.norm()) > 0 else 1.0
    l2 = float(a.pow(2).sum().sqrt())
    print('%s: |g|max %.3e  sample err/scale %.3e  cos %.6f  L2 %.5e vs %.5e' % (prefix, scale, err / scale, cos, l2, st[2]))
    assert torch.isfinite(a).all(), prefix
    assert err <= sample_tol * scale, (prefix, err, scale)
    assert cos >= cos_min, (prefix, cos)
    assert abs(l2 - st[2]) <= l2_rtol * st[2], (prefix, l2, st[2])


def check_summary(z, prefix, t, atol_scale=2e-4, rtol=2e-3):
    """Compare a tensor with the stats/full/sample summary stored by make_golden.summarize."""
    a = t.detach().double().cpu().reshape(-1)
    st = z[prefix + '/stats']
    scale = max(float(st[4]), -float(st[3]), 1e-12)  # max |ref|
    n = a.numel()
    if prefix + '/full' in z.files:
        ref = torch.from_numpy(z[prefix + '/full']).double().reshape(-1)
        report(prefix, a, ref, atol_scale * scale, rtol)
    else:
        ref = torch.from_numpy(z[prefix + '/sample']).double()
        report(prefix + ' (sample)', a[torch.from_numpy(sample_idx(n, ref.numel()))], ref, atol_scale * scale, rtol)
    l2 = float(a.pow(2).sum().sqrt())
    assert abs(l2 - st[2]) <= 1e-3 * st[2] + atol_scale * scale, (prefix, 'L2', l2, st[2])


@pytest.mark.parametrize('name', ['train_1x64', 'train_2x128', 'train_2x96_scaled', 'r50_train_1x128', 'r50_train_2x96'])
def test_train_steps_vs_reference_golden(golden_dir, name):
    """r50_*: the Bottleneck backbone of BASELINE configs[3] (resnet.py:94-159,285-293) assembled by the reference's own
    DBTextModel with the registry entry swapped (tests/golden/make_golden.py)."""
    z = np.load(os.path.join(golden_dir, name + '.npz'))
    n, size, seed, steps = (int(v) for v in z['meta'])
    img, gts = O.synthetic_batch(n, size, seed=seed + 100, img_scale=float(z['img_scale']))
    model = make_model(seed, 'resnet50' if name.startswith('r50') else 'resnet18').train()
    trainer = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
    img, gts = img.to(DEV), gts.to(DEV)
    for it in range(steps):
        preds, losses = trainer.step(img, gts)
        if it == 0:
            report(name + ' preds', preds.cpu(), torch.from_numpy(z['preds']), MAP_ATOL, MAP_RTOL)
            report(name + ' P,T (tight)', preds[:, :2].cpu(), torch.from_numpy(z['preds'][:, :2]), 1e-4, 1e-3)
            for k in [f[len('grad/'):-len('/stats')] for f in z.files if f.startswith('grad/') and f.endswith('/stats')]:
                if k.endswith('.bias') and ('conv.bias' in k or k.endswith(('.0.bias', '.3.bias'))):
                    continue  # conv bias ahead of train-mode BN: analytically zero, reference value is round-off noise
                if name.startswith('r50'):
                    # 53 conv layers: proportionally more ReLU-mask flips than resnet18.  Measured worst cases (round 3):
                    # cosine 0.9983, L2 6.1 % (a one-element bias), single sample 0.27 of |g|max (an 8-element sample of a bias
                    # vector); the principled bound on the Bottleneck backward is test_distance_to_fp64_is_within_the_references_own
                    # (HIP no further from the fp64 gradient than 1.5x the reference's own fp32 run, per tensor)
                    check_grad_summary(z, 'grad/' + k, model.engine.grad_views[k], l2_rtol=0.08, sample_tol=0.35, cos_min=0.997)
                else:
                    check_grad_summary(z, 'grad/' + k, model.engine.grad_views[k])
        tol = ((1e-5, 1e-2, 5e-2) if name.startswith('r50') else (1e-5, 2e-3, 2e-2))[it]
        report('%s losses step %d' % (name, it), losses.cpu().double(), torch.from_numpy(z['losses'][it]), tol, tol)
    sd = model.state_dict()
    if steps == 1:  # after later Adam steps the statistics inherit the chaotic weight differences
        for f in z.files:
            if f.startswith('post/') and f.endswith('/stats'):
                k = f[len('post/'):-len('/stats')]
                if 'running' in k:
                    check_summary(z, 'post/' + k, sd[k], 1e-4, 1e-3)
    assert int(sd['backbone.bn1.num_batches_tracked']) == steps


def test_eval_mode_vs_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, 'eval_2x128.npz'))
    n, size, seed, _ = (int(v) for v in z['meta'])
    img, gts = O.synthetic_batch(n, size, seed=seed + 100)
    model = make_model(seed).eval()
    with torch.no_grad():
        preds = model(img.to(DEV))
        val = DBLoss()(preds, gts.to(DEV))
    assert preds.shape == (n, 2, size, size)
    report('eval preds', preds.cpu(), torch.from_numpy(z['preds']), MAP_ATOL, MAP_RTOL)
    report('eval loss', val.cpu().double().view(1), torch.from_numpy(z['loss']).view(1), 1e-4, 1e-3)


def test_autograd_surface_matches_trainer_and_oracle():
    """The reference call surface (train.py:160-172) with torch.optim.Adam == the fused trainer == oracle."""
    seed, n, size = 9, 2, 64
    img, gts = O.synthetic_batch(n, size, seed=seed)
    sd = O.new_state(seed)
    opt_o = O.AdamState(lr=0.005)
    m1 = make_model(seed).train()
    crit = DBLoss(alpha=1.0, beta=10.0, negative_ratio=3, reduction='mean').to(DEV)
    opt1 = torch.optim.Adam(m1.parameters(), lr=0.005, weight_decay=0, amsgrad=False)
    m2 = make_model(seed).train()
    tr2 = DBTrainer(m2, DBLoss(), FusedAdam(m2, lr=0.005))
    imgd, gtsd = img.to(DEV), gts.to(DEV)
    for it in range(2):
        preds_o, losses_o = O.train_step(sd, opt_o, img, gts)
        preds1 = m1(imgd)
        assert preds1.size(1) == 3
        l1 = crit(preds1, gtsd)
        opt1.zero_grad()
        l1[4].backward()
        if it == 0:
            for k, p in m1.named_parameters():  # .grad on every used parameter (train.py:171)
                assert (p.grad is None) == k.startswith(('backbone.fc', 'backbone.smooth')), k
            g_auto = {k: p.grad.clone() for k, p in m1.named_parameters() if p.grad is not None}
        opt1.step()
        preds2, l2 = tr2.step(imgd, gtsd)
        tol = (1e-5, 2e-3)[it]
        if it == 0:
            report('step 0 preds autograd-vs-oracle', preds1.detach().cpu(), preds_o, MAP_ATOL, MAP_RTOL)
            report('step 0 preds trainer-vs-autograd', preds2.cpu(), preds1.detach().cpu(), 0, 0)
            for k, g in g_auto.items():
                assert torch.equal(g, m2.engine.grad_views[k]), 'autograd and trainer paths disagree on grad ' + k
        report('step %d losses autograd' % it, torch.stack([v.detach() for v in l1]).cpu().double(), torch.tensor(losses_o).double(),
               tol, tol)
        report('step %d losses trainer' % it, l2.cpu().double(), torch.tensor(losses_o).double(), tol, tol)
    for k in ('backbone.fc.weight', 'backbone.smooth.weight'):
        assert dict(m1.named_parameters())[k].grad is None  # dead params
    for k in ('backbone.conv1.weight', 'segmentation_body.conv.0.weight', 'segmentation_head.thresh.6.weight',
              'backbone.layer3.0.bn2.weight'):
        # two Adam steps move every weight by <= 2*lr = 0.01; weights whose ~zero gradient flipped sign differ by up to that
        report_robust('post-step param ' + k, m1.state_dict()[k].cpu(), sd[k], 2.5e-3, 0, 0.97)
        report('post-step param bound ' + k, m1.state_dict()[k].cpu(), sd[k], 0.0201, 0)
        report_robust('post-step param (trainer) ' + k, m2.state_dict()[k].cpu(), m1.state_dict()[k].cpu(), 2.5e-3, 0, 0.97)


def test_cfg1_2x640_vs_reference_golden(golden_dir):
    """BASELINE configs[0]: 2x3x640x640 forward + DBLoss (+ 3 Adam steps) vs the reference's numbers."""
    z = np.load(os.path.join(golden_dir, 'cfg1_2x640.npz'))
    n, size, seed, steps = (int(v) for v in z['meta'])
    img, gts = O.synthetic_batch(n, size, seed=seed + 100)
    model = make_model(seed).train()
    tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
    img, gts = img.to(DEV), gts.to(DEV)
    for it in range(steps):
        preds, losses = tr.step(img, gts)
        if it == 0:
            for c, nm in enumerate('PTB'):
                check_summary(z, 'preds_' + nm, preds[:, c], 1e-3, 1e-2)
            for k in ('backbone.conv1.weight', 'backbone.layer2.0.conv1.weight', 'segmentation_body.conv.0.weight',
                      'segmentation_head.binarize.0.weight', 'segmentation_head.thresh.3.weight',
                      'segmentation_head.binarize.6.weight'):
                check_grad_summary(z, 'grad/' + k, model.engine.grad_views[k])
        tol = (1e-5, 2e-3, 2e-2)[it]
        report('cfg1 losses step %d' % it, losses.cpu().double(), torch.from_numpy(z['losses'][it]), tol, tol)


@pytest.mark.parametrize('math', ['f32', 'bf16x3'])
def test_cfg2_16x640_vs_reference_golden(golden_dir, math):
    """BASELINE configs[1] — the benchmarked workload itself (16x3x640x640, one train step, train.py:160-172) against the
    REFERENCE's own numbers (tests/golden/make_golden.py --only-cfg2): 4096-point samples + L2 norms of the three maps at
    the north_star tolerance, the five losses, six representative gradients, the running statistics after the step.
    Train-mode BatchNorm merges 3200 tile partials per channel here (400 in the 2x640 golden).
    `bf16x3` (fp32 tensors, every product as an exact three-way bf16 split on the bf16 matrix pipe — here through the
    pixel-patch kernels) has to meet the SAME bounds as the exact-fp32 instruction: it is the fp32-accurate alternative line
    of bench.py (`alt_modes.bf16x3`)."""
    z = np.load(os.path.join(golden_dir, 'cfg2_16x640.npz'))
    n, size, seed, steps = (int(v) for v in z['meta'])
    assert (n, size, steps) == (16, 640, 1)
    img, gts = O.synthetic_batch(n, size, seed=seed + 100)
    model = make_model(seed).train()
    model.engine.set_conv_math(math)
    tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
    preds, losses = tr.step(img.to(DEV), gts.to(DEV))
    for c, nm in enumerate('PTB'):
        check_summary(z, 'preds_' + nm, preds[:, c], MAP_ATOL, MAP_RTOL)
    for c, nm in enumerate('PT'):  # and tight: the maps agree far inside the north_star bound
        check_summary(z, 'preds_' + nm, preds[:, c], 1e-4, 1e-3)
    for k in ('backbone.conv1.weight', 'backbone.layer2.0.conv1.weight', 'segmentation_body.conv.0.weight',
              'segmentation_head.binarize.0.weight', 'segmentation_head.thresh.3.weight', 'segmentation_head.binarize.6.weight'):
        check_grad_summary(z, 'grad/' + k, model.engine.grad_views[k])
    report('cfg2 losses', losses.cpu().double(), torch.from_numpy(z['losses'][0]), 1e-5, 1e-5)
    sd = model.state_dict()
    for f in z.files:
        if f.startswith('post/') and f.endswith('/stats') and 'running' in f:
            k = f[len('post/'):-len('/stats')]
            check_summary(z, 'post/' + k, sd[k], 1e-4, 1e-3)


# conv biases ahead of a train-mode BatchNorm: analytically zero gradient (fp64: ~1e-17), the fp32 value is round-off noise
_DEAD_BIAS = ('conv.bias', '.0.bias', '.3.bias')


# (golden, backbone, precision mode): the reference's own distance from fp64 at that precision is the yardstick
_FP64_CASES = [('fp64_2x128', 'resnet18', 'f32'), ('fp64_r50_2x96', 'resnet50', 'f32'), ('fp64_r50_2x96_bn3x02', 'resnet50', 'f32'),
               ('fp64_2x128', 'resnet18', 'f32-direct'), ('fp64_r50_2x96', 'resnet50', 'f32-direct'), ('fp64_r50_2x96_bn3x02', 'resnet50', 'f32-direct'),
               ('fp64_2x128', 'resnet18', 'bf16x3'), ('fp64_2x128', 'resnet18', 'bf16'), ('fp64_r50_2x96_bn3x02', 'resnet50', 'bf16'),
               ('fp64_r50_2x96', 'resnet50', 'bf16'), ('fp64_2x128', 'resnet18', 'bf16c')]


@pytest.mark.parametrize('case,arch,math', _FP64_CASES)
def test_distance_to_fp64_is_within_the_references_own(golden_dir, case, arch, math):
    """DESIGN §4's claim as an assertion, for every precision mode and both block types (BasicBlock: resnet.py:70-91,
    Bottleneck: resnet.py:94-159).  Two implementations at the same precision disagree on the gradients because each
    flips the ReLU mask of different near-zero activations; neither is "right".  Ground truth: the same weights and inputs
    evaluated in DOUBLE (oracle in fp64, pinned to the reference's own .double() run by the golden and
    tests/test_oracle_golden.py).  Yardstick: how far the REFERENCE ITSELF is from that ground truth — in fp32 for the
    'f32' / 'bf16x3' modes, under torch.autocast('cpu', bfloat16) for the native 'bf16' mode (tests/golden/make_golden.py
    case_fp64).  Required of the HIP path:
      per parameter tensor  |g_hip - g64| <= F_t * |g_ref - g64| + 2e-5 |g64|       (F_t = 1.5 fp32 — 2.0 only for the <= 64-element
                            tensors of the Winograd configuration, see below —, 2.0 bf16)
      whole model           |g_hip - g64| <= F_m * |g_ref - g64|                     (F_m = 1.2 fp32, 1.5 bf16)
      maps (strided sample) mean |map_hip - map64| <= 1.5 * reference's + 1e-6; P,T max likewise (+2e-5)
      losses                max_i |l_hip - l64| <= F_l * max_i |l_ref - l64| + 1e-5  (F_l = 1.5 fp32; 3.0 bf16: five scalars of a
                            run whose maps are off by 1e-2..1e-1 on average are single noisy draws).
    '_bn3x02': every bn3 gain scaled by 0.2 (the conditioning of the deep-net bf16 tests; the as-initialised 53-layer net
    amplifies perturbations ~1e3x — measured and stated by the 'fp64_r50_2x96' rows, which run the unconditioned net)."""
    z = np.load(os.path.join(golden_dir, case + '.npz'))
    n, size, seed, _ = (int(v) for v in z['meta'])
    img, gts = O.synthetic_batch(n, size, seed=seed + 100)
    sd = O.new_state(seed, arch)
    gain = float(z['bn3_gain'])
    if gain != 1.0:
        for k in sd:
            if k.endswith('bn3.weight'):
                sd[k] = sd[k] * gain
    _, l64, g64 = O.loss_and_grads(O.to_dtype(sd, torch.float64), img.double(), gts.double())
    assert np.allclose(l64, z['losses_f64'], rtol=1e-10)
    bf16 = math in ('bf16', 'bf16c')  # (bf16c: fp32 tensors, operands rounded to bf16 when staged — same yardstick)
    ref_tag, dist_tag = ('bf16ac', 'refbf16_dist/') if bf16 else ('f32', 'ref32_dist/')
    # fp32: the 3x3 / stride-1 convs of the default path run through Winograd F(2x2,3x3) — fp32 arithmetic whose rounding error constant
    # is about twice the direct sum's.  Whole-model distances stay BELOW the reference's own (measured 0.74x on the unconditioned
    # resnet50, 0.8-1.0x elsewhere); ONE small tensor reaches 1.65x (a 64-element BatchNorm bias of the unconditioned resnet50).  Round 4
    # had widened the per-tensor factor to 2.0 for every tensor of the fp32 mode (advisor finding); it is 1.5 again, as in round 3 —
    # for every tensor of the direct configuration ('f32-direct': Winograd off, the exact k-ordered fmaf chain) and for every tensor
    # of more than 64 elements of the default (Winograd) configuration; only its <= 64-element tensors keep 2.0.
    direct = math == 'f32-direct'
    math = 'f32' if direct else math
    wino = math == 'f32' and not direct
    f_t, f_m = (2.0, 1.5) if bf16 else (1.5, 1.2)
    model = make_model(seed, arch)
    model.load_state_dict(sd)
    model = model.train()
    model.engine.set_conv_math(math)
    if direct:
        model.engine.winograd = model.engine.winograd_wgrad = False
    tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
    preds, losses = tr.step(img.to(DEV), gts.to(DEV))
    # losses
    dl_h = np.abs(losses.cpu().double().numpy() - z['losses_f64']).max()
    dl_r = np.abs(z['losses_' + ref_tag] - z['losses_f64']).max()
    # maps on the golden's strided sample
    p64 = z['preds_f64/sample']
    idx = torch.from_numpy(sample_idx(preds.numel(), p64.size))
    ph = preds.detach().double().cpu().reshape(-1)[idx].numpy()
    ch = (idx.numpy() // (size * size)) % 3
    d_h, d_r = np.abs(ph - p64), np.abs(z['preds_' + ref_tag + '/sample'] - p64)
    print('%s %s %s: losses max|d| hip %.3e ref %.3e; maps mean hip %.3e ref %.3e; P,T max hip %.3e ref %.3e' %
          (case, arch, math, dl_h, dl_r, d_h.mean(), d_r.mean(), d_h[ch < 2].max(), d_r[ch < 2].max()))
    worst, worst_k, tot_h, tot_r, bad = 0.0, None, 0.0, 0.0, []
    for k, g in g64.items():
        if g is None or k.endswith(_DEAD_BIAS):
            continue
        nrm = float(z['g64/' + k + '/norm'])
        d_ref = float(z[dist_tag + k])
        d_hip = float((model.engine.grad_views[k].double().cpu() - g).norm())
        tot_h += d_hip**2
        tot_r += d_ref**2
        ratio = d_hip / (d_ref + 2e-5 * nrm)
        if ratio > worst:
            worst, worst_k = ratio, k
        f_k = 2.0 if (wino and g.numel() <= 64) else f_t
        if d_hip > f_k * d_ref + 2e-5 * nrm:
            bad.append((k, d_hip, d_ref, nrm))
    print('fp64 check: worst d_hip/d_ref %.3f (%s); whole-model |g_hip-g64| %.4e vs reference %.4e (ratio %.3f)' %
          (worst, worst_k, tot_h**0.5, tot_r**0.5, (tot_h / tot_r)**0.5))
    assert dl_h <= (3.0 if bf16 else 1.5) * dl_r + 1e-5
    assert d_h.mean() <= 1.5 * d_r.mean() + 1e-6
    assert d_h[ch < 2].max() <= 1.5 * d_r[ch < 2].max() + 2e-5
    assert not bad, bad
    assert tot_h**0.5 <= f_m * tot_r**0.5


def test_cfg4_r50dcn_800_bs8_f32_and_bf16():
    """BASELINE configs[3] in its own terms: deformable ResNet-50 backbone (resnet.py:94-159, DCN :54-65,111-124) at
    8x3x800x800.  Eval mode (images independent: running-stat BatchNorm): images 0 and 7 of the batch per pixel against the
    CPU oracle; then the 'bf16' conv-math mode on the deformable backbone at the same size (mean error; stated bound);
    then one train step at full size in both modes (size-independent properties; the Bottleneck/DCN train arithmetic
    itself is compared per element at 2x128 above).  DCN sampling: restated DCNv1, parity unpinned against torchvision."""
    seed, arch = 23, 'deformable_resnet50'
    img, gts = O.synthetic_batch(8, 800, seed=seed + 1)
    sd = O.new_state(seed, arch)
    # The procedural fill gives every BatchNorm gain ~1, so each of the 16 Bottleneck residual branches is as strong as its
    # shortcut and the random 53-layer net amplifies a relative perturbation ~1e3x (measured: bf16 operand rounding, 2^-9, comes
    # out as a 50 % error of the FPN features — tools/bf16_dcn_probe.py).  A trained net is shortcut-dominated; the test net
    # gets the standard "small last gain" of residual nets (bn3 gain x 0.2), which leaves the arithmetic under test unchanged.
    for k in sd:
        if k.endswith('.bn3.weight'):
            sd[k] = sd[k] * 0.2
    O.BN_MOMENTUM = 1.0  # running statistics that match the weights (see test_bottleneck_and_deformable_backbones_vs_oracle)
    try:
        with torch.no_grad():
            O.forward(sd, img[[0, 7]], training=True, update_stats=True)
    finally:
        O.BN_MOMENTUM = 0.1
    with torch.no_grad():
        ref = O.forward(sd, img[[0, 7]], training=False)
    model = make_model(seed, arch)
    model.load_state_dict(sd)
    model.eval()
    imgd, gtsd = img.to(DEV), gts.to(DEV)
    with torch.no_grad():
        pe = model(imgd)
    assert pe.shape == (8, 2, 800, 800)
    report('cfg4 f32 eval image 0', pe[0].cpu(), ref[0], MAP_ATOL, MAP_RTOL)
    report('cfg4 f32 eval image 7', pe[7].cpu(), ref[1], MAP_ATOL, MAP_RTOL)
    model.engine.set_conv_math('bf16')  # configs[3] in its own dtype: NATIVE bf16 (storage + MFMA), deformable blocks included
    with torch.no_grad():
        pb = model(imgd)
    assert model.engine.bufs['backbone.layer2.0/cols'].dtype == torch.bfloat16
    err = (pb[[0, 7]].cpu() - ref).abs()
    print('cfg4 bf16 conv math, eval: mean abs err %.3e, max %.3e' % (float(err.mean()), float(err.max())))
    assert torch.isfinite(pb).all() and float(err.mean()) < 2e-2
    tot = {}
    for math in ('f32', 'bf16'):
        m2 = make_model(seed, arch)
        m2.load_state_dict(sd)
        m2.train()
        m2.engine.set_conv_math(math)
        tr = DBTrainer(m2, DBLoss(), FusedAdam(m2, lr=0.005))
        preds, losses = tr.step(imgd, gtsd)
        assert preds.shape == (8, 3, 800, 800) and torch.isfinite(preds).all() and torch.isfinite(losses).all()
        assert float(preds.min()) >= 0 and float(preds.max()) <= 1
        assert torch.allclose(preds[:, 2], torch.sigmoid(50 * (preds[:, 0] - preds[:, 1])), atol=1e-5)
        g = m2.engine.flat_grad
        assert torch.isfinite(g).all() and float(g.abs().max()) > 0
        assert float(m2.engine.grad_views['backbone.layer2.0.conv2_offset.weight'].abs().max()) > 0
        tot[math] = losses.cpu().tolist()
        del m2, tr
        torch.cuda.empty_cache()
    print('cfg4 train losses f32 %s\n                  bf16 %s' % (tot['f32'], tot['bf16']))
    for a, b in zip(tot['f32'], tot['bf16']):  # the stated bf16 bound on the losses (as test_split_bf16_conv_math_modes)
        assert abs(a - b) <= 3e-2 * max(abs(a), 1e-3)


def test_full_size_bs16_properties():
    """BASELINE configs[1] size (16x3x640x640): size-independent properties of the output maps."""
    seed = 0
    img, gts = O.synthetic_batch(16, 640, seed=seed + 100)
    model = make_model(seed).train()
    tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
    img, gts = img.to(DEV), gts.to(DEV)
    preds, losses = tr.step(img, gts)
    assert preds.shape == (16, 3, 640, 640) and torch.isfinite(preds).all() and torch.isfinite(losses).all()
    assert float(preds.min()) >= 0 and float(preds.max()) <= 1
    P, T, B = preds[:, 0], preds[:, 1], preds[:, 2]
    assert torch.allclose(B, torch.sigmoid(50 * (P - T)), atol=1e-5)
    # run-to-run determinism (no atomics anywhere on the path):
    model2 = make_model(seed).train()
    preds2 = model2.engine.forward(img, train=True)
    assert torch.equal(preds, preds2), 'forward is not run-to-run deterministic'
    g = model.engine.flat_grad
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0
    l0 = float(losses[4])
    for _ in range(3):
        _, losses = tr.step(img, gts)
    assert float(losses[4]) < l0, 'loss did not decrease over 4 Adam steps on a fixed batch'


def test_split_bf16_conv_math_modes():
    """'bf16x3' (three-way exact operand split on the bf16 matrix pipe, fp32 accumulate) must meet the SAME north_star tolerance
    as the exact-fp32 path (1e-3 abs / 1e-2 rel on all three maps, losses 1e-5, gradient cosine >= 0.999 against the oracle).
    The 16-bit modes ('bf16', 'bf16c') are NOT judged by absolute bounds any more (round 3 had cosine >= 0.70 / maps 8e-2 here,
    which only hid regressions): test_distance_to_fp64_is_within_the_references_own holds them to the reference's own distance
    from fp64 under bf16 autocast, per tensor; here only the storage contract of the native mode is checked."""
    seed, n, size = 11, 2, 128
    img, gts = O.synthetic_batch(n, size, seed=seed)
    sd = O.new_state(seed)
    model = make_model(seed).train()
    model.engine.set_conv_math('bf16x3')
    tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
    preds, losses = tr.step(img.to(DEV), gts.to(DEV))
    assert preds.dtype == torch.float32
    preds_o, losses_o, grads_o = O.loss_and_grads(sd, img, gts)
    report('bf16x3 maps', preds.cpu(), preds_o, 1e-3, 1e-2)
    report('bf16x3 losses', losses.cpu().double(), torch.tensor(losses_o).double(), 1e-5, 1e-5)
    for k in ('backbone.conv1.weight', 'backbone.layer2.0.conv1.weight', 'segmentation_body.conv.0.weight',
              'segmentation_head.binarize.3.weight', 'segmentation_head.thresh.0.weight', 'backbone.layer4.1.bn2.weight'):
        a, b = model.engine.grad_views[k].cpu().double().flatten(), grads_o[k].double().flatten()
        cos = float(a @ b / (a.norm() * b.norm()))
        print('bf16x3 grad %s: cos %.6f, |g| ratio %.5f' % (k, cos, float(a.norm() / b.norm())))
        assert cos >= 0.999, (k, cos)
    m2 = make_model(seed).train()
    m2.engine.set_conv_math('bf16')  # native storage: 16-bit activations and gradients, fp32 maps / parameters / gradients of parameters
    p2, _ = DBTrainer(m2, DBLoss(), FusedAdam(m2, lr=0.005)).step(img.to(DEV), gts.to(DEV))
    assert p2.dtype == torch.float32
    assert m2.engine.bufs['fpn/z'].dtype == torch.bfloat16 and m2.engine.bufs['backbone.layer1.0/dy1'].dtype == torch.bfloat16
    assert m2.engine.flat.dtype == torch.float32 and m2.engine.flat_grad.dtype == torch.float32


def test_native_bf16_training_and_fp16_inference_full_size():
    """BASELINE configs[2] (bf16, 16x3x640x640 per GPU) and configs[4] (fp16 inference) on the native 16-bit data paths.
    Training: four Adam steps on a fixed batch in bf16 storage next to the fp32 path on the same batch — finite, in range,
    B == sigmoid(50(P-T)) to the map's fp32 evaluation, losses within the stated 4 % of the fp32 path's at step 0, the loss
    decreases, run-to-run bit-reproducible.  Inference: eval-mode forward in fp16 and bf16 storage against the CPU oracle
    (two images, running-statistics BatchNorm); stated bound: mean |err| <= 4e-3, max <= 6e-2 on P,T."""
    seed = 0
    img, gts = O.synthetic_batch(16, 640, seed=seed + 100)
    imgd, gtsd = img.to(DEV), gts.to(DEV)
    ref_model = make_model(seed).train()
    _, l32 = DBTrainer(ref_model, DBLoss(), FusedAdam(ref_model, lr=0.005)).step(imgd, gtsd)
    del ref_model
    runs = []
    for _ in range(2):
        model = make_model(seed).train()
        model.engine.set_conv_math('bf16')
        tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
        preds, losses = tr.step(imgd, gtsd)
        runs.append((preds.clone(), losses.clone(), model.engine.flat_grad.clone()))
    preds, losses, g = runs[0]
    assert all(torch.equal(a, b) for a, b in zip(runs[0], runs[1])), 'bf16 step is not run-to-run deterministic'
    assert preds.shape == (16, 3, 640, 640) and torch.isfinite(preds).all() and torch.isfinite(g).all()
    assert float(preds.min()) >= 0 and float(preds.max()) <= 1
    assert torch.allclose(preds[:, 2], torch.sigmoid(50 * (preds[:, 0] - preds[:, 1])), atol=1e-5)
    print('bf16 losses', losses.cpu().tolist(), 'fp32', l32.cpu().tolist())
    for a, b in zip(losses.cpu().tolist(), l32.cpu().tolist()):
        assert abs(a - b) <= 4e-2 * max(abs(b), 1e-3), (a, b)
    l0 = float(losses[4])
    for _ in range(3):
        _, losses = tr.step(imgd, gtsd)
    assert float(losses[4]) < l0
    del tr, model
    torch.cuda.empty_cache()
    # inference: the oracle's running statistics are calibrated with one momentum-1 train pass so that eval activations are sane
    sd = O.new_state(seed)
    O.BN_MOMENTUM = 1.0
    try:
        with torch.no_grad():
            O.forward(sd, img[:2], training=True, update_stats=True)
    finally:
        O.BN_MOMENTUM = 0.1
    with torch.no_grad():
        ref = O.forward(sd, img[:2], training=False)
    m = make_model(seed)
    m.load_state_dict(sd)
    m.eval()
    for math in ('fp16', 'bf16'):
        m.engine.set_conv_math(math)
        with torch.no_grad():
            pe = m(imgd[:2])
        assert pe.dtype == torch.float32 and pe.shape == (2, 2, 640, 640)
        err = (pe.cpu() - ref).abs()
        print('%s inference: mean |err| %.3e max %.3e' % (math, float(err.mean()), float(err.max())))
        lim = (4e-3, 6e-2) if math == 'fp16' else (2e-2, 2e-1)
        assert float(err.mean()) <= lim[0] and float(err.max()) <= lim[1]
    m.engine.set_conv_math('fp16')
    m.train()
    with pytest.raises(RuntimeError, match='inference'):
        m.engine.forward(imgd[:2], train=True)


def test_cfg5_inference_1280_bs32_vs_oracle():
    """BASELINE configs[4] shape: eval-mode forward at 32x3x1280x1280, in fp32 and on the native fp16 inference path.  In eval mode
    images are independent (running-stat BN), so two of the 32 images are checked per pixel against the CPU oracle; the prob map
    goes to the host exactly as postprocess.py consumes it.  The running statistics are CALIBRATED first (one momentum-1 train pass
    of the oracle over the two checked images, as in test_native_bf16_training_and_fp16_inference_full_size): with the procedurally
    filled statistics eval activations grow to ~1e3 and saturated logits make a max-error bound meaningless — round 3 asserted
    the mean only; now mean AND max are bounded in fp16 (stated bound of the fp16 path: mean <= 4e-3, max <= 6e-2)."""
    seed = 4
    g = torch.Generator().manual_seed(123)
    img = torch.randn(32, 3, 1280, 1280, generator=g)
    sd = O.new_state(seed)
    O.BN_MOMENTUM = 1.0
    try:
        with torch.no_grad():
            O.forward(sd, img[[0, 31]], training=True, update_stats=True)
    finally:
        O.BN_MOMENTUM = 0.1
    model = make_model(seed)
    model.load_state_dict(sd)
    model.eval()
    with torch.no_grad():
        preds = model(img.to(DEV))
        assert preds.shape == (32, 2, 1280, 1280)
        prob = preds[:, 0, :, :].cpu().numpy()  # postprocess.py:33-34,61-62: pred[:, 0], .cpu().numpy()
        ref = O.forward(sd, img[[0, 31]], training=False)
    assert prob.dtype == np.float32 and np.isfinite(prob).all()
    report('cfg5 image 0', preds[0].cpu(), ref[0], MAP_ATOL, MAP_RTOL)
    report('cfg5 image 31', preds[31].cpu(), ref[1], MAP_ATOL, MAP_RTOL)
    # configs[4] in its own dtype: the native fp16 inference path (fp16 activations and weight panels, fp32 accumulate)
    model.engine.set_conv_math('fp16')
    with torch.no_grad():
        preds_f16 = model(img.to(DEV))
    assert preds_f16.dtype == torch.float32 and model.engine.bufs['fpn/z'].dtype == torch.float16
    assert torch.isfinite(preds_f16).all()
    for i, j in ((0, 0), (31, 1)):
        err = (preds_f16[i].cpu() - ref[j]).abs()
        print('cfg5 fp16 inference image %d: mean abs err %.3e, max %.3e' % (i, float(err.mean()), float(err.max())))
        assert float(err.mean()) <= 4e-3 and float(err.max()) <= 6e-2


def test_checkpoint_roundtrip_and_lr_schedulers(tmp_path):
    """train.py:119-139,288-318: state_dict save/load with the reference key set, torch LR schedulers driving FusedAdam."""
    seed = 2
    img, gts = O.synthetic_batch(2, 64, seed=seed)
    model = make_model(seed).train()
    opt = FusedAdam(model, lr=0.005)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 0.5**it)  # any _LRScheduler (e.g. the reference's WarmupPolyLR)
    plateau = torch.optim.lr_scheduler.ReduceLROnPlateau(FusedAdam(make_model(seed)), mode='min', factor=0.1, patience=0)
    tr = DBTrainer(model, DBLoss(), opt)
    tr.step(img.to(DEV), gts.to(DEV))
    sched.step()
    assert abs(opt.param_groups[0]['lr'] - 0.0025) < 1e-12
    plateau.step(1.0); plateau.step(2.0)
    path = str(tmp_path / 'dbnet.pth')
    torch.save(model.state_dict(), path)
    assert os.path.getsize(path) < 80e6  # the flat buffer is stored once, not once per parameter view
    sd = torch.load(path, map_location='cpu')
    assert list(sd.keys()) == [k for k, _, _ in O.state_spec()]
    m2 = DBTextModel()
    m2.load_state_dict(sd)
    m2 = m2.to(DEV).eval()
    model.eval()
    with torch.no_grad():
        a, b = model(img.to(DEV)), m2(img.to(DEV))
    assert torch.equal(a, b)
    assert int(sd['backbone.bn1.num_batches_tracked']) == 1


def test_odd_size_vs_reference_golden(golden_dir):
    """1x3x96x70 against the REFERENCE's own output (tests/golden/odd_1x96x70.npz, make_golden.case_odd): H, W not multiples of 32,
    so the FPN's size-based nearest upsampling (segmentation_body.py:64-76) and the final bilinear(align_corners=True) resample
    (models.py:43-46) are real resamples — dbn_nearest_up_*, dbn_bilinear_fwd/bwd.  Eval maps + loss, train maps, losses,
    gradients, running statistics."""
    z = np.load(os.path.join(golden_dir, 'odd_1x96x70.npz'))
    n, h, w, seed = (int(v) for v in z['meta'])
    img, gts = O.synthetic_batch(n, (h, w), seed=seed + 100)
    model = make_model(seed).eval()
    with torch.no_grad():
        pe = model(img.to(DEV))
        val = DBLoss()(pe, gts.to(DEV))
    assert pe.shape == (n, 2, h, w)
    report('odd-size eval maps vs reference', pe.cpu(), torch.from_numpy(z['eval_preds']), MAP_ATOL, MAP_RTOL)
    # the eval loss of a net with procedurally filled running statistics is ill-conditioned (saturated maps: BCE takes log of
    # values near 0, the loss is ~19): 1e-4 on the maps moves it by 1e-3 relative.  Two checks instead of one loose one: the loss
    # kernel on OUR maps against the oracle's DBLoss on the same maps (tight), and the end-to-end value against the reference's.
    report('odd-size eval loss, HIP DBLoss vs oracle DBLoss on the HIP maps', val.cpu().double().view(1),
           O.db_loss(pe.cpu(), gts).double().view(1), 1e-5, 1e-5)
    report('odd-size eval loss vs reference', val.cpu().double().view(1), torch.from_numpy(z['eval_loss']).view(1), 1e-4, 5e-3)
    model = make_model(seed).train()
    trainer = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
    preds, losses = trainer.step(img.to(DEV), gts.to(DEV))
    assert preds.shape == (n, 3, h, w)
    report('odd-size train maps vs reference', preds.cpu(), torch.from_numpy(z['preds']), MAP_ATOL, MAP_RTOL)
    report('odd-size P,T (tight)', preds[:, :2].cpu(), torch.from_numpy(z['preds'][:, :2]), 1e-4, 1e-3)
    report('odd-size losses vs reference', losses.cpu().double(), torch.from_numpy(z['losses']), 1e-5, 1e-5)
    for k in [f[len('grad/'):-len('/stats')] for f in z.files if f.startswith('grad/') and f.endswith('/stats')]:
        if k.endswith('.bias') and ('conv.bias' in k or k.endswith(('.0.bias', '.3.bias'))):
            continue  # conv bias ahead of train-mode BN: analytically zero, reference value is round-off noise
        # ONE image of 96x70: the maps behind layer2..4 have 108 / 35 / 12 pixels, so a single ReLU-mask flip moves a
        # per-channel gradient element by up to ~5 % of the tensor's largest (measured 5.1 % on layer2.0.bn2.bias, cosine 0.9998)
        # (round 5: 8.3 % on layer2.1.bn2.bias) — the default 5 % everywhere else (round 4: 10 % for every tensor), 10 % on the
        # per-channel BatchNorm tensors of those three stages
        small_map_bn = k.startswith(('backbone.layer2', 'backbone.layer3', 'backbone.layer4')) and ('.bn' in k or 'downsample.1' in k)
        check_grad_summary(z, 'grad/' + k, model.engine.grad_views[k], sample_tol=0.10 if small_map_bn else 0.05)
    sd = model.state_dict()
    for f in z.files:
        if f.startswith('post/') and f.endswith('/stats'):
            check_summary(z, f[:-len('/stats')], sd[f[len('post/'):-len('/stats')]], 1e-4, 1e-3)


@pytest.mark.parametrize('n,h,w', [(1, 96, 70), (2, 70, 90), (1, 33, 47)])
def test_arbitrary_input_sizes(n, h, w):
    """The inference CLIs resize without padding (utils.py:160-175), so H, W are not multiples of 32 there: the FPN's
    size-based nearest upsampling and the final bilinear(align_corners=True) resample (models.py:43-46) become real
    resamples.  Eval and train forward/backward vs the oracle."""
    seed = 8
    g = torch.Generator().manual_seed(5)
    img = torch.randn(n, 3, h, w, generator=g)
    u = torch.rand(4, n, h, w, generator=g)
    gts = torch.stack([(u[0] > 0.9).float(), (u[1] > 0.05).float(), 0.3 + 0.4 * u[2], (u[3] > 0.8).float()])
    model = make_model(seed).eval()
    with torch.no_grad():
        pe = model(img.to(DEV))
        ref_e = O.forward(O.new_state(seed), img, training=False)
    assert pe.shape == (n, 2, h, w)
    report('odd-size eval maps', pe.cpu(), ref_e, MAP_ATOL, MAP_RTOL)
    model.train()
    sd = O.new_state(seed)
    preds_o, losses_o, grads_o = O.loss_and_grads(sd, img, gts)
    preds = model(img.to(DEV))
    losses = DBLoss()(preds, gts.to(DEV))
    losses[4].backward()
    report('odd-size train maps (P,T)', preds[:, :2].detach().cpu(), preds_o[:, :2], MAP_ATOL, MAP_RTOL)
    report('odd-size losses', torch.stack([v.detach() for v in losses]).cpu().double(), torch.tensor(losses_o).double(), 1e-4, 1e-3)
    for k in ('backbone.conv1.weight', 'segmentation_body.conv.0.weight', 'segmentation_head.thresh.3.weight'):
        a, b = dict(model.named_parameters())[k].grad.cpu().double().flatten(), grads_o[k].double().flatten()
        cos = float(a @ b / (a.norm() * b.norm()))
        print('odd-size grad %s cos %.6f' % (k, cos))
        assert cos > 0.995, (k, cos)


@pytest.mark.parametrize('arch,n,size', [('deformable_resnet50', 2, 128), ('deformable_resnet18', 2, 64), ('resnet50', 1, 160)])
def test_bottleneck_and_deformable_backbones_vs_oracle(arch, n, size):
    """BASELINE configs[3] backbone family (resnet.py:94-159, DCN :54-65,111-124; SURVEY A4'): train step (maps, losses,
    gradients incl. conv2_offset) and eval forward vs the CPU oracle.  DCN arithmetic: restated DCNv1 (parity unpinned
    against torchvision, see oracle.deform_conv2d); the Bottleneck net itself is pinned by the r50_* goldens."""
    seed = 21
    img, gts = O.synthetic_batch(n, size, seed=seed + 1)
    sd = O.new_state(seed, arch)
    # eval mode needs running statistics that match the weights (the procedural ones let activations of a 50-layer net
    # explode, and a deformable conv then samples chaotically): calibrate them with one momentum-1 train-mode pass
    O.BN_MOMENTUM = 1.0
    try:
        with torch.no_grad():
            O.forward(sd, img, training=True, update_stats=True)
    finally:
        O.BN_MOMENTUM = 0.1
    model = make_model(seed, arch)
    model.load_state_dict(sd)
    model.eval()
    assert list(model.state_dict().keys()) == list(sd.keys())
    with torch.no_grad():
        pe = model(img.to(DEV))
    report(arch + ' eval preds', pe.cpu(), O.forward(sd, img, training=False), MAP_ATOL, MAP_RTOL)
    model.train()
    trainer = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
    preds, losses = trainer.step(img.to(DEV), gts.to(DEV))
    preds_o, losses_o, grads_o = O.loss_and_grads(sd, img, gts)
    report(arch + ' preds', preds.cpu(), preds_o, MAP_ATOL, MAP_RTOL)
    report(arch + ' losses', losses.cpu().double(), torch.tensor(losses_o).double(), 1e-4, 1e-4)
    worst = 1.0
    for k, g in grads_o.items():
        if g is None or (k.endswith('.bias') and ('conv.bias' in k or k.endswith(('.0.bias', '.3.bias')))):
            continue
        a, b = model.engine.grad_views[k].double().cpu().reshape(-1), g.double().reshape(-1)
        if float(b.norm()) == 0:
            continue
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
        rel = float((a - b).norm() / b.norm())
        worst = min(worst, cos)
        assert cos >= 0.997 and rel <= 0.08, (k, cos, rel)  # (measured worst: cosine 0.9988; see also the fp64-distance test)
    print('%s: worst gradient cosine %.6f' % (arch, worst))
    if 'deformable' in arch:
        k = 'backbone.layer2.0.conv2_offset.weight'
        assert float(model.engine.grad_views[k].abs().max()) > 0
    assert worst >= (0.995 if 'resnet50' in arch else 0.9999), worst
    for k, v in model.state_dict().items():  # running statistics after one train step
        if 'running' in k:
            assert torch.allclose(v.cpu(), sd[k], atol=1e-4, rtol=1e-3), k


@pytest.mark.parametrize('arch', ['deformable_resnet18', 'deformable_resnet50', 'resnet50'])
def test_native_bf16_on_bottleneck_and_deformable_backbones(arch):
    """Native bf16 storage on the configs[3] backbone family at a size the oracle runs in seconds: one train step (maps at the
    stated bf16 bound, losses, gradient direction incl. conv2_offset) and eval forward vs the fp32 CPU oracle.  The test net
    gets the small last BatchNorm gain of residual nets (see test_cfg4_r50dcn_800_bs8_f32_and_bf16)."""
    seed, n, size = 21, 2, 128
    img, gts = O.synthetic_batch(n, size, seed=seed + 1)
    sd = O.new_state(seed, arch)
    for k in sd:
        if k.endswith('.bn3.weight'):
            sd[k] = sd[k] * 0.2
    model = make_model(seed, arch)
    model.load_state_dict(sd)
    model.train()
    model.engine.set_conv_math('bf16')
    tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
    preds, losses = tr.step(img.to(DEV), gts.to(DEV))
    preds_o, losses_o, grads_o = O.loss_and_grads(sd, img, gts)
    # stated bf16 bound on P,T for these deeper / deformable nets: 99.9 % of the pixels within 8e-2 (+8e-2 rel), mean |err| <= 2e-2
    report_robust(arch + ' bf16 maps', preds[:, :2].cpu(), preds_o[:, :2], 8e-2, 8e-2, 0.999)
    assert float((preds[:, :2].cpu() - preds_o[:, :2]).abs().mean()) <= 2e-2
    report(arch + ' bf16 losses', losses.cpu().double(), torch.tensor(losses_o).double(), 4e-2, 4e-2)
    keys = ['backbone.conv1.weight', 'segmentation_body.conv.0.weight', 'segmentation_head.binarize.3.weight']
    if 'deformable' in arch:
        keys += ['backbone.layer2.0.conv2_offset.weight', 'backbone.layer3.0.conv2.weight']
        assert model.engine.bufs['backbone.layer2.0/cols'].dtype == torch.bfloat16
    for k in keys:
        a, b = model.engine.grad_views[k].cpu().double().flatten(), grads_o[k].double().flatten()
        cos = float(a @ b / (a.norm() * b.norm()))
        print('%s bf16 grad %s: cos %.5f' % (arch, k, cos))
        # (measured 0.74-0.98; the reference's own bf16-autocast run reaches 0.75-0.99 on the same kind of net, see the
        # 'refbf16_cos' entries of tests/golden/fp64_r50_2x96_bn3x02.npz — the deformable nets have no reference yardstick:
        # torchvision is absent)
        assert cos >= 0.7, (k, cos)


def test_two_stream_step_is_bit_reproducible():
    """The step's second HIP stream (weight gradients, threshold branch, FPN laterals) changes WHEN kernels run, not what they
    compute: gradients and updated parameters are bit-identical to the single-stream schedule and from run to run
    (no atomics on the ResNet-18 path; every reduction has a fixed order)."""
    seed = 9
    img, gts = O.synthetic_batch(2, 96, seed=seed)
    outs = []
    for overlap in (True, False, True):
        model = make_model(seed).train()
        model.engine.overlap_wgrad = overlap
        trainer = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
        for _ in range(2):
            preds, losses = trainer.step(img.to(DEV), gts.to(DEV))
        torch.cuda.synchronize()
        outs.append((model.engine.flat_grad.clone(), model.engine.flat.clone(), preds.clone(), losses.clone()))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


@pytest.mark.parametrize('math', ['f32', 'bf16'])
def test_deformable_backbone_step_is_bit_reproducible(math):
    """The deformable convs' sampling adjoint accumulates in 64-bit fixed point since round 3 (rounds 1-2: float atomics, the
    library's only kernel whose summation order was not fixed): a train step of the DCN backbone (resnet.py:54-65,111-124) with
    NON-ZERO learned offsets gives bit-identical gradients, parameters and maps from run to run."""
    seed = 13
    img, gts = O.synthetic_batch(2, 96, seed=seed)
    sd = O.new_state(seed, 'deformable_resnet18')
    g = torch.Generator().manual_seed(1)
    for k in sd:  # the reference initialises conv2_offset to zero (resnet.py:204-208): give it something to sample off-grid
        if 'conv2_offset' in k:
            sd[k] = torch.randn(sd[k].shape, generator=g) * (0.05 if k.endswith('weight') else 0.7)
    outs = []
    for _ in range(3):
        model = make_model(seed, 'deformable_resnet18')
        model.load_state_dict(sd)
        model = model.train()
        model.engine.set_conv_math(math)
        tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
        for _ in range(2):
            preds, losses = tr.step(img.to(DEV), gts.to(DEV))
        torch.cuda.synchronize()
        outs.append((model.engine.flat_grad.clone(), model.engine.flat.clone(), preds.clone(), losses.clone()))
    assert float(outs[0][0].abs().max()) > 0 and torch.isfinite(outs[0][0]).all()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


@pytest.mark.parametrize('math', ['f32', 'bf16'])
def test_deformable_backbone_gather_adjoint_equals_the_fixed_point_scatter(math):
    """Round 5's sampling adjoint (a per-pixel gather in plain fp32, engine.dcn_gather) against round 3's fixed-point scatter on the whole
    DCN backbone (resnet.py:54-65,111-124) with non-zero learned offsets: the first step's gradients agree to the fp32 rounding of the sums
    (bf16: to the one extra rounding of dx / doffset to storage), maps and losses likewise."""
    seed = 13
    img, gts = O.synthetic_batch(2, 96, seed=seed)
    sd = O.new_state(seed, 'deformable_resnet18')
    g = torch.Generator().manual_seed(1)
    for k in sd:
        if 'conv2_offset' in k:
            sd[k] = torch.randn(sd[k].shape, generator=g) * (0.05 if k.endswith('weight') else 0.7)
    outs = []
    for gather in (False, True):
        model = make_model(seed, 'deformable_resnet18')
        model.load_state_dict(sd)
        model = model.train()
        model.engine.set_conv_math(math)
        model.engine.dcn_gather = gather
        tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
        preds, losses = tr.step(img.to(DEV), gts.to(DEV))
        torch.cuda.synchronize()
        outs.append((model.engine.flat_grad.clone().double(), preds.clone().double(), losses.clone().double()))
    (g0, p0, l0), (g1, p1, l1) = outs
    assert torch.equal(p0, p1) and torch.equal(l0, l1)  # (the forward pass is the same code)
    tol = 2e-2 if math == 'bf16' else 1e-4
    rel = float((g0 - g1).norm() / g0.norm())
    print('gather vs scatter, %s: relative gradient distance %.3e' % (math, rel))
    assert rel <= tol, rel
    # per parameter tensor (a small tensor must not hide behind the large ones)
    views0 = model.engine.grad_views
    worst = 0.0
    for k, v in views0.items():
        n = v.numel()
        a = g0[v.storage_offset():v.storage_offset() + n] if v.storage_offset() + n <= g0.numel() else None
        if a is None:
            continue
        b = g1[v.storage_offset():v.storage_offset() + n]
        if float(a.norm()) > 0:
            worst = max(worst, float((a - b).norm() / a.norm()))
    print('worst per-tensor relative distance %.3e' % worst)
    assert worst <= (10 * tol if math == 'bf16' else 20 * tol), worst


def test_deformable_adjoint_is_chosen_from_the_previous_steps_offsets():
    """engine._dcn_read_back / _dcn_send_back: the sampling adjoint of a deformable layer is the gather while the PREVIOUS step's max |offset|
    of that layer is at most engine.dcn_gather_max_offset, the fixed-point scatter beyond — decided without a synchronisation inside the
    step, deterministically: two runs agree bit for bit, and the maxima read back equal the offsets the forward pass produced."""
    seed = 13
    img, gts = O.synthetic_batch(2, 96, seed=seed)
    sd = O.new_state(seed, 'deformable_resnet18')
    g = torch.Generator().manual_seed(1)
    names = [k for k in sd if 'conv2_offset' in k]
    for k in names:  # half of the deformable layers get offsets of tens of pixels, the others stay below one
        big = (names.index(k) // 2) % 2 == 0
        sd[k] = torch.randn(sd[k].shape, generator=g) * (0.02 if k.endswith('weight') else (25.0 if big else 0.4))
    runs = []
    for _ in range(2):
        model = make_model(seed, 'deformable_resnet18')
        model.load_state_dict(sd)
        model = model.train()
        eng = model.engine
        tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=1e-4))
        tr.step(img.to(DEV), gts.to(DEV))
        first = dict(eng.dcn_forms)
        offs = {k[:-len('/offset')]: float(t.float()[..., :18].abs().max()) for k, t in eng.bufs.items() if k.endswith('/offset')}
        preds, losses = tr.step(img.to(DEV), gts.to(DEV))
        torch.cuda.synchronize()
        second = {k: eng.dcn_forms[k] - first[k] for k in first}
        runs.append((eng.flat_grad.clone(), eng.flat.clone(), preds.clone(), first, second, dict(eng._dcn_E), offs))
    _, _, _, first, second, seen, offs = runs[0]
    nl = len(offs)
    assert nl >= 4 and first == {'gather': nl, 'scatter': 0}, (first, nl)  # nothing is known in the first step: the gather
    nbig = sum(1 for v in offs.values() if v > model.engine.dcn_gather_max_offset)
    assert 0 < nbig < nl and second == {'gather': nl - nbig, 'scatter': nbig}, (second, nbig, offs)
    for name, v in offs.items():  # what the second step read back is the first step's maximum, exactly
        assert seen[name] == pytest.approx(v, rel=0, abs=0), (name, seen[name], v)
    for a, b in zip(runs[0][:3], runs[1][:3]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('math', ['f32', 'bf16'])
def test_graph_captured_step_is_bit_identical_to_the_eager_step(math):
    """DBTrainer.use_graph: forward + DBLoss + backward replayed as ONE hipGraph launch (two-stream fork / join captured with it),
    gradient exchange and Adam outside.  Five steps over changing batches — two eager warm-up steps, the capturing step, two
    replays — must leave parameters, BatchNorm buffers (incl. num_batches_tracked), maps and losses bit-identical to five
    eager steps, and an eval-mode forward afterwards must see the updated weights (the host-side panel stamps are kept valid)."""
    seed = 9
    batches = [O.synthetic_batch(2, 96, seed=seed + i) for i in range(3)]

    def run(use_graph):
        model = make_model(seed).train()
        model.engine.set_conv_math(math)
        trainer = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
        trainer.use_graph = use_graph
        rec = []
        for it in range(5):
            img, gts = batches[it % 3]
            preds, losses = trainer.step(img.to(DEV), gts.to(DEV))
            rec.append((preds.clone(), losses.clone()))
        assert (trainer._graph is not None and trainer._graph['graph'] is not None) == use_graph
        model.eval()
        with torch.no_grad():
            pe = model(batches[0][0].to(DEV)).clone()
        torch.cuda.synchronize()
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        return rec, pe, sd, model.engine.flat_grad.clone()

    def same(a, b):
        (rec0, pe0, sd0, g0), (rec1, pe1, sd1, g1) = a, b
        for it, ((p0, l0), (p1, l1)) in enumerate(zip(rec0, rec1)):
            if not (torch.equal(p0, p1) and torch.equal(l0, l1)):
                return 'step %d differs between the eager and the graph-captured step' % it
        if not (torch.equal(pe0, pe1) and torch.equal(g0, g1)):
            return 'eval forward / gradients differ'
        for k in sd0:
            if not torch.equal(sd0[k], sd1[k]):
                return k
        return None

    eager = run(False)
    graph = run(True)
    # one run each, hard fail (round 3 retried up to three times and downgraded mismatches to a warning; the graph's input copy was
    # then skipped on an (id, _version, data_ptr) key that a freed temporary's successor can reproduce — train.DBTrainer._graph_step)
    assert same(eager, graph) is None, same(eager, graph)
    assert same(eager, run(False)) is None, 'the eager step did not reproduce itself'
    assert int(graph[2]['backbone.bn1.num_batches_tracked']) == 5


def test_fit_and_evaluate_epoch_loop(tmp_path):
    """train.py:146-318: epochs over a loader of reference-style batch dicts, poly/plateau schedulers,
    eval-mode validation with the pixel metric, the reference's best-checkpoint rule and the final checkpoint."""
    from db_text_minimal_amd.train import evaluate, fit
    seed = 3

    def loader(n_batches, base):
        out = []
        for i in range(n_batches):
            img, gts = O.synthetic_batch(2, 64, seed=base + i)
            out.append({'img': img, 'prob_map': gts[0], 'supervision_mask': gts[1], 'thresh_map': gts[2], 'text_area_map': gts[3]})
        return out

    train_loader, test_loader = loader(3, 100), loader(2, 200)
    model = make_model(seed).train()
    opt = FusedAdam(model, lr=0.005)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode='min', factor=0.5, patience=0)
    best, last = str(tmp_path / 'best.pth'), str(tmp_path / 'last.pth')
    hist = fit(model, DBLoss(), opt, train_loader, test_loader, epochs=3, scheduler=sched, lrs_mode='reduce', thresh=0.3,
               best_cp_path=best, last_cp_path=last, device=DEV)
    assert len(hist) == 3 and hist[-1]['global_steps'] == 9
    assert all(np.isfinite(h['train_loss']) and np.isfinite(h['test_loss']) for h in hist)
    assert hist[-1]['train_loss'] < hist[0]['train_loss']  # three epochs on three fixed batches: the loss goes down
    assert hist[0].get('saved_best') and os.path.exists(best) and os.path.exists(last)
    assert set(hist[0]['test_score']) == {'Overall Acc', 'Mean Acc', 'FreqW Acc', 'Mean IoU'}
    # the saved checkpoint is the reference's wire format and reproduces the evaluation
    m2 = DBTextModel()
    m2.load_state_dict(torch.load(last, map_location='cpu'))
    m2 = m2.to(DEV)
    l1, s1 = evaluate(model, DBLoss(), test_loader, device=DEV)
    l2, s2 = evaluate(m2, DBLoss(), test_loader, device=DEV)
    assert l1 == l2 and s1 == s2 and abs(l1 - hist[-1]['test_loss']) < 1e-6
    # the oracle agrees with evaluate() on the final weights
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    ref = 0.0
    for b in test_loader:
        p = O.forward(sd, b['img'], training=False)
        ref += float(O.db_loss(p, torch.stack([b['prob_map'], b['supervision_mask'], b['thresh_map'], b['text_area_map']])))
    assert abs(ref / len(test_loader) - l1) < 2e-3 * max(1.0, abs(l1))


@pytest.mark.parametrize('math,arch,n,size', [('f32', 'resnet18', 2, 96), ('bf16', 'resnet18', 2, 96), ('bf16x3', 'resnet18', 1, 64),
                                              ('f32', 'resnet18', 1, (96, 70)), ('f32', 'deformable_resnet18', 1, 64),
                                              ('bf16', 'resnet50', 1, 64)])
def test_results_do_not_depend_on_uninitialised_memory(math, arch, n, size):
    """Every buffer the engine allocates (activations, gradients, weight panels, slabs, partial rows, scratch) comes from
    torch.empty: recycled allocator blocks holding whatever the previous owner left.  A kernel that reads an element no kernel of
    this step has written — a partial row of a tile that does not exist, a panel's padding, a slab gap — makes the step depend on
    that history, which shows as a once-in-a-while run-to-run mismatch on some boxes (round 3's unexplained transients).  Here the
    history is made explicit: three training steps with all fresh buffers pre-filled with NaN, with finite noise (different per
    buffer), and left as allocated must produce the same bits (train.py:160-172 is the sequence)."""
    from db_text_minimal_amd import engine as engine_mod
    seed = 13
    img, gts = O.synthetic_batch(n, size, seed=seed)
    img, gts = img.to(DEV), gts.to(DEV)
    sd = O.new_state(seed, arch)

    def run(poison):
        engine_mod.POISON = poison
        try:
            model = DBTextModel() if arch == 'resnet18' else DBTextModel(arch)
            model.load_state_dict(sd)
            model = model.to(DEV).train()
            model.engine.set_conv_math(math)
            tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
            for _ in range(3):
                preds, losses = tr.step(img, gts)
            torch.cuda.synchronize()
            return preds.clone(), losses.clone(), model.engine.flat_grad.clone(), model.engine.flat.clone()
        finally:
            engine_mod.POISON = ''

    base = run('')
    assert torch.isfinite(base[2]).all() and torch.isfinite(base[0]).all()
    for poison in ('nan', 'rand', 'rand'):
        got = run(poison)
        for name, a, b in zip(('preds', 'losses', 'gradients', 'parameters'), base, got):
            bad = int((a != b).sum()) if a.shape == b.shape else -1
            assert torch.equal(a, b), ('%s differ (%d elements) with fresh buffers pre-filled by %r: the step reads memory it has not '
                                       'written' % (name, bad, poison))


def test_fused_adam_refuses_gradient_accumulation():
    """torch.optim.Adam would SUM the gradients of two backward passes without a zero_grad() in between; the fused path's flat
    gradient buffer holds only the last pass — FusedAdam.step() raises instead of silently applying half of the accumulation."""
    seed = 3
    img, gts = O.synthetic_batch(1, 64, seed=seed)
    model = make_model(seed).train()
    opt = FusedAdam(model, lr=0.005)
    crit = DBLoss()
    for _ in range(2):
        crit(model(img.to(DEV)), gts.to(DEV))[4].backward()
    with pytest.raises(RuntimeError, match='backward passes'):
        opt.step()
    opt.zero_grad()
    crit(model(img.to(DEV)), gts.to(DEV))[4].backward()
    opt.step()  # one pass since zero_grad(): fine
    crit(model(img.to(DEV)), gts.to(DEV))[4].backward()
    opt.step()  # step() also clears the count (the reference calls zero_grad() first anyway, train.py:169-172)


@pytest.mark.parametrize('math,arch', [('f32', 'resnet18'), ('bf16', 'resnet18'), ('f32', 'resnet50')])
def test_grouped_slab_reduction_is_bit_identical_to_per_layer_launches(math, arch):
    """engine.defer_wgrad_reduce: the weight gradients' slab reductions run as one grouped launch per gradient stage
    (dbn_wgrad_reduce_many over a device job table) instead of one small launch behind every matrix kernel.  Same sums in the same
    order: after two training steps gradients and parameters must equal the per-layer form bit for bit — also through the
    bucketed-exchange hook points, which flush the pending reductions before a bucket is announced."""
    seed = 6
    img, gts = O.synthetic_batch(2, 96, seed=seed)
    img, gts = img.to(DEV), gts.to(DEV)
    outs = []
    for defer in (False, True):
        model = make_model(seed, arch).train()
        model.engine.set_conv_math(math)
        model.engine.defer_wgrad_reduce = defer
        tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
        for _ in range(2):
            tr.step(img, gts)
        torch.cuda.synchronize()
        assert bool(model.engine._reduce_tables) == defer and not model.engine._reduce_pending
        outs.append((model.engine.flat_grad.clone(), model.engine.flat.clone()))
    assert float(outs[0][0].abs().max()) > 0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize('size,train', [((160, 160), True), ((96, 70), True), ((128, 160), False)])
def test_apply_on_load_batchnorm_relu_is_bit_identical(size, train):
    """engine.apply_on_load (exact-fp32 mode): the BatchNorm + ReLU in front of a Winograd conv (bn1 of every BasicBlock,
    resnet.py:77-80; the FPN output's BatchNorm, segmentation_body.py:60-61 -> segmentation_head.py:24-25,64-68) is applied while the
    conv and its weight-gradient kernel stage their patches — the activation tensor ('/z1', 'fpn/z') is never written.  Same
    arithmetic as bn_apply: predictions, losses, gradients and parameters after two steps must equal the materialised form bit for bit
    (odd sizes: ragged patches and the consecutive-tile form of the small maps)."""
    seed = 9
    img, gts = O.synthetic_batch(2, size, seed=seed)
    img, gts = img.to(DEV), gts.to(DEV)
    outs = []
    for on in (False, True):
        model = make_model(seed, 'resnet18')
        model.engine.apply_on_load = on
        if train:
            model.train()
            tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
            for _ in range(2):
                preds, losses = tr.step(img, gts)
            torch.cuda.synchronize()
            outs.append((preds.clone(), losses.clone(), model.engine.flat_grad.clone(), model.engine.flat.clone()))
        else:
            model.eval()
            model.engine.fold_eval_bn = False  # (round 5: the default eval path folds the BatchNorm into the weights and writes the
            # activations from the conv epilogues; apply-on-load belongs to the unfolded conv -> coefficients -> bn_apply chain)
            with torch.no_grad():
                outs.append((model(img).clone(), ))
        written = sorted(k for k in model.engine.bufs if k.endswith('/z1') or k == 'fpn/z')
        # (off: the 8 blocks' z1 and fpn/z exist; on: none of the maps the Winograd kernels take — the smallest maps of a small input stay direct)
        assert (len(written) == 9) if not on else ('backbone.layer1.0/z1' not in written and 'fpn/z' not in written), written
    for a, b in zip(*outs):
        assert float(a.abs().max()) > 0 and torch.equal(a, b)


@pytest.mark.gpu
def test_pyramid_conv_on_a_winograd_level_0():
    """engine.fpn_level0_winograd (opt-in): the FPN output conv's level 0 — the plain 3x3 conv of p2 (segmentation_body.py:55-61,82-87) —
    through the Winograd kernel, levels 1-3 added onto it by dbn_pyramid_conv_from_t(first_level = 1), against the one-launch pyramid
    conv: same sum in another order — predictions and losses to 2e-5; every gradient tensor to 1e-2 of its scale after one step (measured
    1.5e-3 at the stem: a rounding-level change of the FPN output flips ReLU masks of near-zero activations in the unconditioned random-init
    net, which changes gradients discretely — the same sensitivity the fp64 yardstick of the other tests is built around)."""
    seed = 11
    img, gts = O.synthetic_batch(2, 160, seed=seed)
    img, gts = img.to(DEV), gts.to(DEV)
    outs = []
    for on in (False, True):
        model = make_model(seed, 'resnet18').train()
        model.engine.fpn_level0_winograd = on
        tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
        preds, losses = tr.step(img, gts)
        torch.cuda.synchronize()
        assert (('segmentation_body.conv.0#lv0', 'winograd', 64, 1) in model.engine.packs) == on
        outs.append((preds.clone(), losses.clone(), {k: v.clone() for k, v in model.engine.grad_views.items()}))
    (p0, l0, g0), (p1, l1, g1) = outs
    assert float((p0 - p1).abs().max()) <= 2e-5 and float((l0 - l1).abs().max()) <= 2e-5 * float(l0.abs().max())
    for k in g0:
        scale = float(g0[k].abs().max())
        assert float((g0[k] - g1[k]).abs().max()) <= 1e-2 * scale + 1e-7, k  # (+ 1e-7: biases in front of a BatchNorm have zero gradient — rounding noise of 1e-9)


@pytest.mark.parametrize('math,n,h,w', [('bf16', 2, 128, 128), ('fp16', 2, 160, 128), ('fp16', 1, 96, 70), ('bf16', 1, 200, 136)])
def test_inference_fusions_on_16bit_storage_equal_the_unfused_launches(math, n, h, w):
    """Round 5, inference on 16-bit storage: the stem conv + BatchNorm + ReLU + max-pool in one launch (engine.stem16_pool), the head's ConvT ->
    BatchNorm -> ReLU -> ConvT -> sigmoid of both branches in one launch (engine.head16), the pointwise kernel of the c2 lateral (engine.pw16)
    and the projection shortcut on the second stream — against the same model with every one of them switched off: the same function up to
    where the 16-bit roundings fall (mean difference of the maps bounded like two 16-bit evaluations of one net), both within the oracle's
    bounds; sizes that are no multiples of 32 take the real resamples behind the fused head (models.py:43-46); run-to-run bit identity."""
    seed = 23
    img, _ = O.synthetic_batch(n, (h, w), seed=seed)
    sd = O.new_state(seed)
    model = make_model(seed)
    model.engine.set_conv_math(math)
    model.eval()
    ref = O.forward(sd, img, training=False, update_stats=False)
    eng = model.engine

    def run(on):
        eng.stem16_pool = eng.head16 = eng.pw16 = eng.eval_downsample_beside = on
        with torch.no_grad():
            return model(img.to(DEV)).clone()
    a, b = run(True), run(False)
    assert a.shape == (n, 2, h, w) and torch.isfinite(a).all()
    d = (a - b).abs()
    print('fused vs unfused (%s %dx%d): max %.3e mean %.3e' % (math, h, w, float(d.max()), float(d.mean())))
    assert float(d.mean()) <= 4e-3
    for t in (a, b):
        assert float((t.cpu() - ref).abs().mean()) <= 4e-3
    assert torch.equal(run(True), a)


@pytest.mark.parametrize('math,arch,n,h,w', [('f32', 'resnet18', 2, 128, 128), ('f32', 'resnet18', 1, 96, 70), ('bf16', 'resnet18', 2, 128, 128),
                                             ('fp16', 'resnet18', 2, 160, 128), ('f32', 'resnet50', 1, 96, 96), ('f32', 'deformable_resnet18', 1, 96, 96)])
def test_eval_with_folded_batchnorm_equals_the_unfolded_chain(math, arch, n, h, w):
    """Round 5: in eval mode every conv -> BatchNorm -> (+ residual) -> ReLU chain runs as ONE launch on weights with the running statistics
    folded in (engine.fold_eval_bn; basic.py:32-36, resnet.py:70-91,135-159, segmentation_body.py:55-61 under model.eval()).  Same function
    as the unfolded chain (conv -> coefficients -> bn_apply, engine.fold_eval_bn = False) up to the rounding of w * scale — fp32: 2e-5 on the
    maps; 16-bit storage: the folded weights are rounded to the storage type (the unfolded chain rounds the raw weights and applies the
    scale in fp32), bounded like two 16-bit runs of one net — and both within the usual bounds of the oracle.  Also after a train step
    (the running statistics change through raw pointers: the folded tensors must follow)."""
    seed = 21
    img, gts = O.synthetic_batch(n, (h, w), seed=seed)
    sd = O.new_state(seed, arch)
    model = make_model(seed, arch)
    model.engine.set_conv_math(math)
    ref = O.forward(sd, img, training=False, update_stats=False)

    def run(fold):
        model.engine.fold_eval_bn = fold
        model.eval()
        with torch.no_grad():
            return model(img.to(DEV)).clone()
    a, b = run(True), run(False)
    assert a.shape == (n, 2, h, w)

    def close(tag, u, v, trained=False):
        # the eval maps of a net with procedurally filled running statistics are saturated sigmoids: rounding-level differences in the
        # logits (w * scale is rounded once more in the folded form) move single pixels by 1e-4 (resnet18) ... 4e-3 (the unconditioned
        # 53-layer nets, which amplify perturbations ~1e3x); in 16-bit storage single pixels flip outright, as between any
        # two 16-bit evaluations of one net — there the MEAN is what is bounded
        d = (u - v).abs()
        print('%s: max %.3e mean %.3e' % (tag, float(d.max()), float(d.mean())))
        if math == 'f32':
            # (after a train step the single worst pixel sits wherever that step's weights put a logit near 0: 9.4e-4 with round 4's pool
            # backward, 1.37e-3 with the recorded argmax, means 1.3e-7 both — the mean bound is the one that says "same function")
            assert float(d.max()) <= ((3e-3 if trained else 1e-3) if arch == 'resnet18' else 1e-2), tag
            assert float(d.mean()) <= (1e-5 if arch == 'resnet18' else 2e-4), tag
        else:
            # (two 16-bit evaluations of one net: the mean moves with the summation order of the conv kernels — 3.55e-3 with the
            # pixel-patch kernels, 4.03e-3 with round 6's weight-resident ones on the same weights; either form is within 4e-3 of the ORACLE below)
            assert float(d.mean()) <= 5e-3, tag
    close('folded vs unfolded eval maps (%s, %s)' % (math, arch), a.cpu(), b.cpu())
    assert torch.equal(run(True), a)  # (cached folded tensors: same bits)
    # ... and the folded path against the oracle at the bounds the unfolded one is held to (fp32 resnet18: north_star; else DESIGN section 4)
    if math == 'f32' and arch == 'resnet18':
        report('folded eval maps vs oracle', a.cpu(), ref, MAP_ATOL, MAP_RTOL)
    else:
        d = (a.cpu() - ref).abs()
        assert float(d.mean()) <= 4e-3 and (math != 'f32' or float(d.max()) <= 1e-2), (float(d.mean()), float(d.max()))
    if math in ('f32', 'bf16'):  # a train step moves the running statistics and the weights: the folded tensors are re-made
        model.train()
        tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
        tr.step(img.to(DEV), gts.to(DEV))
        a2, b2 = run(True), run(False)
        close('folded vs unfolded after a train step', a2.cpu(), b2.cpu(), trained=True)
        assert not torch.equal(a2, a)


def test_stem_backward_from_the_recorded_argmax_against_the_two_pass_form():
    """engine.pool_argmax (late in round 5): the stem's pool records its argmax in the forward pass and the backward goes from dpool to the
    gradient at the conv output in one pass (dbn_maxpool_bn_backward_t) — against round 4's max-pool backward + BatchNorm backward pair.
    fp32, a random image (no exact ties): the same gradients to rounding.  bf16: the pair gives a window's gradient to EVERY position that
    ties with the maximum (frequent with 8 mantissa bits), the new form to the first one as nn.MaxPool2d does — so the two differ, and the
    new form must not be farther from the fp32 gradients than the old one.  Forward results are the same code."""
    seed = 21
    img, gts = O.synthetic_batch(2, 96, seed=seed)
    keys = ('backbone.conv1.weight', 'backbone.bn1.weight', 'backbone.bn1.bias')
    res = {}
    for math in ('f32', 'bf16'):
        for arg in (False, True):
            model = make_model(seed, 'resnet18').train()
            model.engine.set_conv_math(math)
            model.engine.pool_argmax = arg
            tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
            preds, losses = tr.step(img.to(DEV), gts.to(DEV))
            torch.cuda.synchronize()
            eng = model.engine
            stem = {k: eng.grad_views[k].clone().double() for k in keys}
            res[(math, arg)] = (eng.flat_grad.clone().double(), preds.clone().double(), losses.clone().double(), stem)
    for math in ('f32', 'bf16'):
        assert torch.equal(res[(math, False)][1], res[(math, True)][1]) and torch.equal(res[(math, False)][2], res[(math, True)][2])
    g0, g1 = res[('f32', False)][0], res[('f32', True)][0]
    assert float((g0 - g1).norm() / g0.norm()) <= 1e-5
    for k in keys:
        a, b = res[('f32', False)][3][k], res[('f32', True)][3][k]
        rel = float((a - b).norm() / a.norm())
        print('f32 %s: two-pass vs recorded argmax %.3e' % (k, rel))
        assert rel <= 1e-5, (k, rel)
        ref = b
        d_old = float((res[('bf16', False)][3][k] - ref).norm() / ref.norm())
        d_new = float((res[('bf16', True)][3][k] - ref).norm() / ref.norm())
        print('bf16 %s: distance to the fp32 gradient %.3e (every tie) / %.3e (first maximum)' % (k, d_old, d_new))
        assert d_new <= 1.1 * d_old + 1e-3, (k, d_old, d_new)
