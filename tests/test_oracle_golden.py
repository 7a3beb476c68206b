"""CPU: pin the oracle (oracle/dbnet_oracle.py) against golden vectors produced by the REFERENCE
itself (tests/golden/make_golden.py imported /root/reference/src/{models,losses}.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import dbnet_oracle as O


def sample_idx(numel, k=256):
    if numel <= k:
        return np.arange(numel)
    return (np.arange(k, dtype=np.int64) * (numel // k)) + (numel // (2 * k))


def test_state_spec_matches_reference_layout():
    spec = O.state_spec()
    assert len(spec) == 211
    n_params = sum(int(np.prod(s)) for k, s, kind in spec if kind not in ('bn_rm', 'bn_rv', 'bn_nbt'))
    assert n_params == 13306922
    live = sum(int(np.prod(s)) for k, s, kind in spec if kind not in ('bn_rm', 'bn_rv', 'bn_nbt', 'dead'))
    assert live == 12269378


@pytest.mark.parametrize('name', ['train_1x64', 'train_2x128', 'train_2x96_scaled', 'r50_train_1x128', 'r50_train_2x96'])
def test_oracle_train_steps_match_reference(golden_dir, name):
    torch.set_num_threads(8)
    z = np.load(os.path.join(golden_dir, name + '.npz'))
    n, size, seed, steps = (int(v) for v in z['meta'])
    img, gts = O.synthetic_batch(n, size, seed=seed + 100, img_scale=float(z['img_scale']))
    sd = O.new_state(seed, 'resnet50' if name.startswith('r50') else 'resnet18')
    opt = O.AdamState(lr=0.005)
    for it in range(steps):
        if it == 0:
            preds, losses, grads = O.loss_and_grads(sd, img, gts)
            assert np.abs(preds.numpy() - z['preds']).max() < 1e-6
            for f in z.files:
                if f.startswith('grad/') and f.endswith('/stats'):
                    k = f[5:-6]
                    a = grads[k].double().reshape(-1).numpy()
                    st = z[f]
                    assert abs(np.sqrt((a * a).sum()) - st[2]) <= 1e-5 * st[2] + 1e-12, k
                    key = 'grad/' + k + ('/full' if 'grad/' + k + '/full' in z.files else '/sample')
                    ref = z[key].reshape(-1)
                    got = a if key.endswith('full') else a[sample_idx(a.size)]
                    assert np.abs(got - ref).max() <= 1e-5 * max(abs(st[3]), abs(st[4])) + 1e-12, k
            with torch.no_grad():
                opt.step(sd, grads)
        else:
            preds, losses = O.train_step(sd, opt, img, gts)
        assert np.allclose(losses, z['losses'][it], rtol=2e-4, atol=1e-6), (it, losses, z['losses'][it])


def test_oracle_cfg2_full_size_golden_is_self_consistent(golden_dir):
    """BASELINE configs[1] golden (the reference at 16x3x640x640, one train step): too large to re-run in the CPU suite;
    the fixture's own invariants are checked here, the HIP path is compared with it in tests/test_model_gpu.py."""
    z = np.load(os.path.join(golden_dir, 'cfg2_16x640.npz'))
    assert [int(v) for v in z['meta']] == [16, 640, 16, 1]
    assert z['preds_P/sample'].shape == (4096, ) and z['losses'].shape == (1, 5)
    P, T, B = (z['preds_%s/sample' % c].astype(np.float64) for c in 'PTB')
    assert np.abs(B - 1.0 / (1.0 + np.exp(-50.0 * (P - T)))).max() < 2e-5  # segmentation_head.py:106-108
    l = z['losses'][0]
    assert abs(l[3] - (l[0] + 10.0 * l[1])) < 1e-5 and abs(l[4] - (l[2] + l[3])) < 1e-5  # losses.py:129-136


@pytest.mark.parametrize('case,arch', [('fp64_2x128', 'resnet18'), ('fp64_r50_2x96', 'resnet50'), ('fp64_r50_2x96_bn3x02', 'resnet50')])
def test_oracle_fp64_matches_reference_in_double(golden_dir, case, arch):
    """The fp64 fixtures (reference .double() next to its own fp32 and bf16-autocast runs, tests/golden/make_golden.py:case_fp64)
    pin the oracle evaluated in double, which the GPU gradient tests use as the ground truth — BasicBlock and Bottleneck nets."""
    torch.set_num_threads(8)
    z = np.load(os.path.join(golden_dir, case + '.npz'))
    n, size, seed, _ = (int(v) for v in z['meta'])
    img, gts = O.synthetic_batch(n, size, seed=seed + 100)
    sd = O.new_state(seed, arch)
    if float(z['bn3_gain']) != 1.0:
        for k in sd:
            if k.endswith('bn3.weight'):
                sd[k] = sd[k] * float(z['bn3_gain'])
    # the autocast leg is a yardstick, not something the oracle restates: it must at least be the noisier one
    assert sum(float(z['refbf16_dist/' + k[4:-5]])**2 for k in z.files if k.startswith('g64/') and k.endswith('/norm')) > \
        sum(float(z['ref32_dist/' + k[4:-5]])**2 for k in z.files if k.startswith('g64/') and k.endswith('/norm'))
    _, l64, g64 = O.loss_and_grads(O.to_dtype(sd, torch.float64), img.double(), gts.double())
    _, l32, g32 = O.loss_and_grads(sd, img, gts)
    assert np.allclose(l64, z['losses_f64'], rtol=1e-12) and np.allclose(l32, z['losses_f32'], rtol=1e-6)
    for k, g in g64.items():
        if g is None:
            continue
        a = g.reshape(-1).numpy()
        nrm = float(z['g64/' + k + '/norm'])
        assert abs(np.sqrt((a * a).sum()) - nrm) <= 1e-9 * nrm + 1e-300, k
        assert np.abs(a[sample_idx(a.size)] - z['g64/' + k + '/sample']).max() <= 1e-9 * np.abs(a).max() + 1e-300, k
        d = float((g32[k].double() - g).norm())
        assert abs(d - float(z['ref32_dist/' + k])) <= 1e-3 * float(z['ref32_dist/' + k]) + 1e-12 * nrm, k


def test_oracle_eval_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, 'eval_2x128.npz'))
    n, size, seed, _ = (int(v) for v in z['meta'])
    img, gts = O.synthetic_batch(n, size, seed=seed + 100)
    with torch.no_grad():
        preds = O.forward(O.new_state(seed), img, training=False)
        val = O.db_loss(preds, gts)
    assert preds.shape[1] == 2 and np.abs(preds.numpy() - z['preds']).max() < 1e-6
    assert abs(float(val) - float(z['loss'])) < 1e-6


def test_oracle_odd_size_matches_reference(golden_dir):
    """1x3x96x70 — not a multiple of 32: the FPN's size-based nearest upsampling and the final bilinear(align_corners=True)
    resample are real resamples (models.py:43-46, segmentation_body.py:64-76).  Eval forward + loss and one train step's maps,
    losses, gradients and running statistics of the oracle against the reference's own (make_golden.case_odd)."""
    z = np.load(os.path.join(golden_dir, 'odd_1x96x70.npz'))
    n, h, w, seed = (int(v) for v in z['meta'])
    img, gts = O.synthetic_batch(n, (h, w), seed=seed + 100)
    with torch.no_grad():
        pe = O.forward(O.new_state(seed), img, training=False)
        val = O.db_loss(pe, gts)
    assert pe.shape == (n, 2, h, w) and np.abs(pe.numpy() - z['eval_preds']).max() < 1e-6
    assert abs(float(val) - float(z['eval_loss'])) < 1e-5 * abs(float(z['eval_loss']))
    sd = O.new_state(seed)
    preds, losses, grads = O.loss_and_grads(sd, img, gts)
    assert preds.shape == (n, 3, h, w) and np.abs(preds.numpy() - z['preds']).max() < 1e-6
    assert np.allclose(losses, z['losses'], rtol=1e-5, atol=1e-7)
    for f in z.files:
        if f.startswith('grad/') and f.endswith('/stats'):
            k = f[5:-6]
            a = grads[k].double().reshape(-1).numpy()
            st = z[f]
            assert abs(np.sqrt((a * a).sum()) - st[2]) <= 1e-5 * st[2] + 1e-12, k
            key = 'grad/' + k + ('/full' if 'grad/' + k + '/full' in z.files else '/sample')
            got = a if key.endswith('full') else a[sample_idx(a.size)]
            assert np.abs(got - z[key].reshape(-1)).max() <= 1e-5 * max(abs(st[3]), abs(st[4])) + 1e-12, k
        if f.startswith('post/') and f.endswith('/stats'):
            k = f[5:-6]
            a = sd[k].double().reshape(-1).numpy()
            assert abs(np.sqrt((a * a).sum()) - z[f][2]) <= 1e-5 * z[f][2] + 1e-9, k


@pytest.mark.parametrize('tag', ['default', 'eval2ch', 'no_positive', 'all_masked', 'neg_limited', 'saturated', 'alpha_beta',
                                 'reduction_none', 'reduction_sum', 'fractional_mask', 'fractional_gt_and_mask',
                                 'fractional_mask_topk_selects', 'fractional_mask_sum'])
def test_oracle_loss_known_answers(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, 'loss_kats.npz'))
    preds = torch.from_numpy(z[tag + '/preds']).requires_grad_(True)
    gts = torch.from_numpy(z[tag + '/gts'])
    kw = {}
    if tag == 'alpha_beta':
        kw = dict(alpha=5.0, beta=2.0, negative_ratio=1)
    if tag.startswith('reduction_'):
        kw = dict(reduction=tag[len('reduction_'):])
    if tag.startswith('fractional_mask_'):
        kw = dict(negative_ratio=2, reduction='sum' if tag.endswith('_sum') else 'mean')
    res = O.db_loss(preds, gts, **kw)
    res5 = res if isinstance(res, tuple) else (res, )
    assert np.allclose([float(v) for v in res5], z[tag + '/losses'], rtol=1e-6, atol=1e-7)
    res5[-1].backward()
    assert np.allclose(preds.grad.numpy(), z[tag + '/dpreds'], rtol=1e-5, atol=1e-9)
    if tag.startswith('fractional'):
        # non-binary maps: the closed form is NOT the reference's value (the product refuses such maps or takes the literal form)
        cf = O.db_loss_closed_form(preds.detach(), gts, **kw)
        assert abs(cf[0] - z[tag + '/losses'][0]) > 1e-3 * abs(z[tag + '/losses'][0]), (cf, z[tag + '/losses'])
    elif tag not in ('reduction_none', ):
        # closed form evaluated by the HIP kernel == the literal reference formula (binary maps)
        cf = O.db_loss_closed_form(preds.detach(), gts, **kw)
        ref = z[tag + '/losses']
        cf = cf if len(ref) == 5 else cf[-1:]
        assert np.allclose(cf, ref, rtol=2e-5, atol=1e-6), (cf, ref)


def test_oracle_adam_matches_torch_optim():
    g = torch.Generator().manual_seed(0)
    p = torch.randn(1000, generator=g)
    q = torch.nn.Parameter(p.clone())
    topt = torch.optim.Adam([q], lr=0.005, weight_decay=0, amsgrad=False)
    sd, opt = {'w': p.clone()}, O.AdamState(lr=0.005)
    for i in range(3):
        gr = torch.randn(1000, generator=g) * 10**(-i)
        q.grad = gr.clone()
        topt.step()
        opt.step(sd, {'w': gr})
    assert torch.allclose(sd['w'], q.data, rtol=1e-6, atol=1e-7)


def test_dp_golden_is_mean_of_shard_grads(golden_dir):
    """SURVEY.md §8e pin, oracle side: averaging per-shard oracle grads reproduces the reference's."""
    z = np.load(os.path.join(golden_dir, 'dp_2x1x128.npz'))
    _, size, seed, _ = (int(v) for v in z['meta'])
    img, gts = O.synthetic_batch(2, size, seed=seed + 100)
    acc = None
    for r in range(2):
        _, losses, gr = O.loss_and_grads(O.new_state(seed), img[r:r + 1], gts[:, r:r + 1])
        assert abs(losses[4] - float(z['loss_rank%d' % r])) < 1e-5
        acc = gr if acc is None else {k: acc[k] + gr[k] for k in gr}
    for k in ('backbone.conv1.weight', 'segmentation_head.thresh.6.weight', 'backbone.layer4.1.bn2.weight'):
        a = (acc[k] / 2).double().reshape(-1).numpy()
        st = z['grad/' + k + '/stats']
        assert abs(np.sqrt((a * a).sum()) - st[2]) <= 1e-5 * st[2]


def test_oracle_pixel_metrics_match_reference(golden_dir):
    """text_metrics.cal_text_score / RunningScore restated in the oracle vs the reference's own output."""
    z = np.load(os.path.join(golden_dir, 'pixel_metrics.npz'))
    hist = np.zeros((2, 2))
    for step in range(2):
        P, G, M = (torch.from_numpy(z['step%d/%s' % (step, k)]) for k in 'PGM')
        hist += O.pixel_confusion(P, G, M, 0.25)
        assert np.array_equal(hist, z['step%d/hist' % step])
        sc = O.scores_from_confusion(hist)
        got = [sc[k] for k in ('Overall Acc', 'Mean Acc', 'FreqW Acc', 'Mean IoU')]
        assert np.allclose(got, z['step%d/scores' % step], rtol=1e-12)


def test_state_specs_of_the_bottleneck_and_deformable_backbones():
    """resnet.py:285-306: resnet50 has 161 parameter tensors / 25.6 M backbone+FPN+head parameters; the deformable variants add
    conv2_offset (18 channels, with bias) to every block of layers 2-4 only (resnet.py:176-191)."""
    r50 = O.state_spec('resnet50')
    assert len(r50) == 409 and O.backbone_out_channels('resnet50') == [256, 512, 1024, 2048]
    d50 = dict((k, s) for k, s, _ in O.state_spec('deformable_resnet50'))
    offs = [k for k in d50 if 'conv2_offset.weight' in k]
    assert len(offs) == 4 + 6 + 3 and not any(k.startswith('backbone.layer1.') for k in offs)
    assert d50['backbone.layer2.0.conv2_offset.weight'] == (18, 128, 3, 3) and d50['backbone.layer2.0.conv2.weight'] == (128, 128, 3, 3)
    d18 = [k for k, _, _ in O.state_spec('deformable_resnet18') if 'conv2_offset.bias' in k]
    assert len(d18) == 6
    for arch in O.ARCHS:
        assert O.arch_of(O.new_state(0, arch)) == arch


def test_deform_conv_oracle_properties():
    """The restated DCNv1 sampling (torchvision is absent: parity unpinned) — pinned by what must hold for any correct
    implementation: zero offsets == F.conv2d (how the reference initialises it, resnet.py:204-208); integer offsets ==
    shifted input; half-pixel offsets == mean of the two neighbours; autograd gradients == finite differences."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 9, 11, generator=g, dtype=torch.double)
    w = torch.randn(7, 5, 3, 3, generator=g, dtype=torch.double)
    for stride in (1, 2):
        ref = F.conv2d(x, w, None, stride, 1)
        off = torch.zeros(2, 18, ref.shape[2], ref.shape[3], dtype=torch.double)
        assert (O.deform_conv2d(x, off, w, stride, 1) - ref).abs().max() < 1e-12
    off = torch.zeros(2, 18, 9, 11, dtype=torch.double)
    off[:, 0::2] = 1.0
    shifted = F.conv2d(F.pad(x, (0, 0, 0, 1))[:, :, 1:], w, None, 1, 1)
    assert (O.deform_conv2d(x, off, w, 1, 1) - shifted)[:, :, 1:].abs().max() < 1e-12
    off[:, 0::2] = 0.5
    half = 0.5 * (F.conv2d(x, w, None, 1, 1) + shifted)
    assert (O.deform_conv2d(x, off, w, 1, 1) - half)[:, :, 1:-1].abs().max() < 1e-12
    off[:, 0::2] = -40.0  # everything sampled outside the image
    assert O.deform_conv2d(x, off, w, 1, 1).abs().max() == 0
    x = torch.randn(1, 2, 5, 6, generator=g, dtype=torch.double, requires_grad=True)
    w = torch.randn(3, 2, 3, 3, generator=g, dtype=torch.double, requires_grad=True)
    off = (torch.rand(1, 18, 5, 6, generator=g, dtype=torch.double) * 1.6 - 0.8 + 0.013).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda a, b, c: O.deform_conv2d(a, b, c, 1, 1), (x, off, w), eps=1e-6, atol=1e-5)


@pytest.mark.parametrize('arch', sorted(O.ARCHS))
def test_fixture_inputs_equal_the_oracles_generators(golden_dir, arch):
    """tests/golden/fixture_inputs.py regenerates a fixture's inputs (procedural weights, synthetic batch) without importing the
    oracle — bench.py's parity gate uses it before its timed region.  Its copy of the generators must equal the oracle's (the
    ones make_golden.py applied to the reference) bit for bit, for every architecture's key set."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('fixture_inputs', os.path.join(golden_dir, 'fixture_inputs.py'))
    fx = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fx)
    ref = O.new_state(16, arch)
    mine = {k: torch.zeros_like(v) for k, v in ref.items()}
    fx.procedural_fill(mine, 16)
    kinds = {k: kind for k, _, kind in O.state_spec(arch)}
    for k, v in ref.items():
        assert fx.kind_of(k, mine) == kinds[k], k
        assert torch.equal(v, mine[k]), k
    for (n, size, seed, scale) in ((2, 32, 116, 1.0), (1, (24, 40), 5, 3.0)):
        a, b = O.synthetic_batch(n, size, seed=seed, img_scale=scale), fx.synthetic_batch(n, size, seed=seed, img_scale=scale)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert (fx.sample_idx(10**6, 4096) == sample_idx(10**6, 4096)).all()
