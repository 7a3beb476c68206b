"""-m gpu: every kernel of libdbnet_hip.so, called through the C ABI, against the
plain fp32 CPU PyTorch op it replaces (same seeded inputs).  Tolerances: the
igemm kernels use exact-f32 MFMA (k-ordered fmaf chains), so they agree with
the CPU result to fp32 round-off (1e-4 abs / 1e-4 rel on O(1..10) values)."""
import pytest
import torch
import torch.nn.functional as F

from gpu_util import DEV, L, igemm, nchw, nhwc, pack, reduce_ws, report, rnd, stream, wgrad
from db_text_minimal_amd import _lib

pytestmark = pytest.mark.gpu


def pad_c(x, c):
    if x.shape[1] == c:
        return x
    return torch.cat([x, torch.zeros(x.shape[0], c - x.shape[1], *x.shape[2:])], 1)


CONV_CASES = [
    # N, Cin, Cout, k, s, p, H, W
    (2, 64, 64, 3, 1, 1, 16, 12),
    (1, 64, 128, 3, 2, 1, 18, 14),
    (2, 64, 128, 1, 2, 0, 16, 16),
    (2, 3, 64, 7, 2, 3, 32, 40),
    (1, 256, 256, 3, 1, 1, 12, 12),
    (3, 128, 64, 1, 1, 0, 9, 7),
    (1, 512, 512, 3, 1, 1, 2, 2),
]


@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('tile', [0, 1, 2, 3, 4])
def test_conv_forward(case, tile):
    N, Ci, Co, k, s, p, H, W = case
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, k, k, seed=2, scale=(2.0 / (Ci * k * k))**0.5)
    b = rnd(Co, seed=3)
    ref = F.conv2d(x, w, b, s, p)
    Ho, Wo = ref.shape[2:]
    xs = nhwc(pad_c(x, (Ci + 3) // 4 * 4))
    y = torch.full((N, Ho, Wo, Co), float('nan'), device=DEV)
    igemm(xs, pack(w, 0), b.to(DEV), y, k, s, p, 0, 0, tile)
    report('conv fwd %s tile %d' % (case, tile), nchw(y), ref, 1e-4, 1e-4)
    # accumulate flag, no bias
    y2 = torch.ones((N, Ho, Wo, Co), device=DEV)
    igemm(xs, pack(w, 0), None, y2, k, s, p, 0, 1, tile)
    report('conv fwd acc %s' % (case, ), nchw(y2), F.conv2d(x, w, None, s, p) + 1, 1e-4, 1e-4)


DGRAD_CASES = [(2, 64, 64, 3, 1, 1, 16, 12), (1, 64, 128, 3, 2, 1, 18, 14), (2, 64, 128, 1, 2, 0, 16, 16), (1, 64, 64, 3, 2, 1, 17, 13),
               (1, 256, 256, 3, 1, 1, 12, 12), (1, 128, 256, 3, 2, 1, 8, 8), (1, 256, 512, 1, 2, 0, 4, 4)]


@pytest.mark.parametrize('case', DGRAD_CASES)
@pytest.mark.parametrize('tile', [0, 4])
def test_conv_dgrad(case, tile):
    N, Ci, Co, k, s, p, H, W = case
    x = rnd(N, Ci, H, W, seed=1).requires_grad_(True)
    w = rnd(Co, Ci, k, k, seed=2, scale=(2.0 / (Ci * k * k))**0.5)
    y = F.conv2d(x, w, None, s, p)
    dy = rnd(*y.shape, seed=4)
    (dx_ref, ) = torch.autograd.grad(y, x, dy)
    dx = torch.full((N, H, W, Ci), float('nan'), device=DEV)
    igemm(nhwc(dy), pack(w, 1, s), None, dx, k, s, p, 1, 0, tile)
    report('conv dgrad %s tile %d' % (case, tile), nchw(dx), dx_ref, 1e-4, 1e-4)
    dx2 = torch.ones((N, H, W, Ci), device=DEV)  # accumulate form
    igemm(nhwc(dy), pack(w, 1, s), None, dx2, k, s, p, 1, 1, tile)
    report('conv dgrad acc %s' % (case, ), nchw(dx2), dx_ref + 1, 1e-4, 1e-4)


WGRAD_CASES = DGRAD_CASES + [(2, 3, 64, 7, 2, 3, 32, 40), (4, 64, 64, 3, 1, 1, 40, 40), (1, 512, 512, 3, 1, 1, 2, 2)]


@pytest.mark.parametrize('case', WGRAD_CASES)
def test_conv_wgrad(case):
    N, Ci, Co, k, s, p, H, W = case
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, k, k, seed=2).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = rnd(*y.shape, seed=4)
    (dw_ref, ) = torch.autograd.grad(y, w, dy)
    g = wgrad(nhwc(dy), nhwc(pad_c(x, (Ci + 3) // 4 * 4)), Co, Ci, k, s, p)
    scale = float(dw_ref.abs().max())
    report('conv wgrad %s' % (case, ), g.cpu(), dw_ref, 2e-5 * scale + 1e-5, 1e-4)


ROW_CASES = [(2, 64, 64, 3, 1, 1, 8, 16), (1, 64, 128, 3, 2, 1, 10, 32), (2, 3, 64, 7, 2, 3, 12, 64), (3, 64, 256, 1, 1, 0, 5, 48),
             (2, 128, 64, 3, 1, 1, 6, 32), (5, 64, 64, 3, 1, 1, 16, 16), (1, 64, 128, 1, 2, 0, 8, 32), (2, 64, 64, 3, 1, 0, 6, 18),
             # 8 x 2 and 4 x 4 pixel blocks
             (2, 64, 64, 3, 1, 1, 6, 24), (3, 128, 128, 3, 1, 1, 8, 20), (1, 64, 128, 3, 2, 1, 8, 48), (2, 3, 64, 7, 2, 3, 16, 40),
             (5, 256, 64, 1, 1, 0, 4, 12), (1, 64, 64, 3, 2, 1, 24, 24), (4, 64, 64, 3, 1, 1, 40, 40), (2, 256, 256, 3, 1, 1, 20, 20)]


@pytest.mark.parametrize('ns', [0, 3])
@pytest.mark.parametrize('case', ROW_CASES)
def test_conv_wgrad_whole_row_addressing(case, ns):
    """Output maps that tile into 16 x 1, 8 x 2 or 4 x 4 pixel blocks: the k-tiles are such blocks and the kernel addresses them
    with loop-invariant per-thread offsets + a scalar offset (wgrad_f32_kernel<..., ROW = 1>).  Equal to autograd's weight gradient;
    with 16 x 1 blocks the pixels are visited in the order of the general gather (dbn_set_wgrad_variant(3)): bit-identical to it."""
    N, Ci, Co, k, s, p, H, W = case
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, k, k, seed=2).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    Ho, Wo = y.shape[2:]
    assert Wo % 16 == 0 or (Wo % 8 == 0 and Ho % 2 == 0) or (Wo % 4 == 0 and Ho % 4 == 0)
    dy = rnd(*y.shape, seed=4)
    (dw_ref, ) = torch.autograd.grad(y, w, dy)
    xs, dys = nhwc(pad_c(x, (Ci + 3) // 4 * 4)), nhwc(dy)
    g = wgrad(dys, xs, Co, Ci, k, s, p, 1.0, ns)
    try:
        _lib.check(L().dbn_set_wgrad_variant(3), 'variant')
        g_general = wgrad(dys, xs, Co, Ci, k, s, p, 1.0, ns)
    finally:
        L().dbn_set_wgrad_variant(0)
    scale = float(dw_ref.abs().max())
    report('conv wgrad (row) ns=%d %s' % (ns, case), g.cpu(), dw_ref, 2e-5 * scale + 1e-5, 1e-4)
    report('conv wgrad (general) ns=%d %s' % (ns, case), g_general.cpu(), dw_ref, 2e-5 * scale + 1e-5, 1e-4)
    if Wo % 16 == 0:
        assert torch.equal(g, g_general)
    assert torch.equal(g, wgrad(dys, xs, Co, Ci, k, s, p, 1.0, ns))  # run to run


@pytest.mark.parametrize('shape', [(2, 64, 64, 8, 6), (1, 64, 64, 16, 16)])
def test_conv_transpose(shape):
    N, Ci, Co, H, W = shape
    x = rnd(N, Ci, H, W, seed=1).requires_grad_(True)
    w = rnd(Ci, Co, 2, 2, seed=2, scale=0.1).requires_grad_(True)
    b = rnd(Co, seed=3)
    ref = F.conv_transpose2d(x, w, b, 2)
    dy = rnd(*ref.shape, seed=5)
    dx_ref, dw_ref = torch.autograd.grad(ref, (x, w), dy)
    xs = nhwc(x.detach())
    y = torch.full((N, 2 * H, 2 * W, Co), float('nan'), device=DEV)
    igemm(xs, pack(w.detach(), 1, 2), b.to(DEV), y, 2, 2, 0, 1)
    report('convT fwd', nchw(y), ref, 1e-4, 1e-4)
    dys = nhwc(dy)
    dx = torch.full((N, H, W, Ci), float('nan'), device=DEV)
    igemm(dys, pack(w.detach(), 0), None, dx, 2, 2, 0, 0)
    report('convT dgrad', nchw(dx), dx_ref, 1e-4, 1e-4)
    g = wgrad(xs, dys, Ci, Co, 2, 2, 0)
    report('convT wgrad', g.cpu(), dw_ref, 1e-4, 1e-4)


@pytest.mark.parametrize('with_bn', [0, 1])
@pytest.mark.parametrize('shape', [(2, 64, 64, 8, 6), (3, 64, 128, 20, 13), (1, 16, 64, 5, 7), (2, 32, 64, 16, 16), (1, 48, 192, 9, 11),
                                   (16, 64, 64, 40, 40)])
def test_conv_transpose_2x2_kernel(shape, with_bn):
    """ConvTranspose2d(2x2, stride 2) forward in exact fp32 runs as convt2x2_f32_kernel (input tile resident in LDS, the four parity
    classes walked inside the workgroup): equals F.conv_transpose2d and — the same products summed in the same order — is
    BIT-IDENTICAL to the general parity-class launch (dbn_set_convt_kernel(0)); with the fused BatchNorm statistics (one partial
    row per workgroup over its 4 x 128 output pixels) the coefficients / running statistics equal F.batch_norm's.  Ragged tiles
    (pixels % 128 != 0), several channel tiles, every supported Cin."""
    N, Ci, Co, H, W = shape
    x = rnd(N, Ci, H, W, seed=1) * 2 + 0.3
    w = rnd(Ci, Co, 2, 2, seed=2, scale=0.1)
    b = rnd(Co, seed=3) * 2
    ref = F.conv_transpose2d(x, w, b, 2)
    xs, wp, bd = nhwc(x), pack(w, 1, 2), b.to(DEV)
    d = lambda t: t.clone().to(DEV)
    gamma, beta = rnd(Co, seed=4) * 0.3 + 1, rnd(Co, seed=5)
    rm, rv = rnd(Co, seed=6), rnd(Co, seed=7).abs() + 0.5
    outs = []
    for on in (1, 0):
        old = L().dbn_set_convt_kernel(on)
        try:
            y = torch.full((N, 2 * H, 2 * W, Co), float('nan'), device=DEV)
            if with_bn:
                g_, b_, rm_, rv_ = d(gamma), d(beta), d(rm), d(rv)
                sc, sh, mu, rs = (torch.full((Co, ), float('nan'), device=DEV) for _ in range(4))
                ws = torch.empty(L().dbn_conv_bn_ws_floats(N, 2 * H, 2 * W, Co, 1, 2), device=DEV)
                _lib.check(L().dbn_conv_bn_f32(xs.data_ptr(), wp.data_ptr(), bd.data_ptr(), y.data_ptr(), N, H, W, Ci, 2 * H, 2 * W, Co, 2, 2,
                                               2, 0, 1, 0, 0, 0, g_.data_ptr(), b_.data_ptr(), 1e-5, 0.1, rm_.data_ptr(), rv_.data_ptr(),
                                               sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), ws.data_ptr(), stream()), 'conv_bn')
                outs.append((y, sc, sh, mu, rs, rm_, rv_))
            else:
                igemm(xs, wp, bd, y, 2, 2, 0, 1)
                outs.append((y, ))
        finally:
            L().dbn_set_convt_kernel(old)
    report('convT 2x2 kernel', nchw(outs[0][0]), ref, 1e-4, 1e-4)
    assert torch.equal(outs[0][0], outs[1][0])
    if with_bn:
        rm_ref, rv_ref = rm.clone(), rv.clone()
        z_ref = F.batch_norm(ref, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
        y, sc, sh, mu, rs, rm_, rv_ = outs[0]
        report('convT+bn running_mean', rm_.cpu(), rm_ref, 1e-5, 1e-5)
        report('convT+bn running_var', rv_.cpu(), rv_ref, 1e-5, 2e-5)
        report('convT+bn normalised output', nchw(y) * sc.cpu().view(1, Co, 1, 1) + sh.cpu().view(1, Co, 1, 1), z_ref, 2e-5, 1e-4)
        for nm, a, g in zip(('scale', 'shift', 'mean', 'rstd'), outs[0][1:5], outs[1][1:5]):  # other partial rows, same statistics
            report('convT+bn %s vs general launch' % nm, a.cpu(), g.cpu(), 2e-6 * float(g.abs().max()) + 1e-7, 1e-5)


@pytest.mark.parametrize('C,N,H,W', [(64, 2, 12, 10), (128, 1, 7, 5), (256, 2, 6, 6), (512, 3, 2, 2), (64, 4, 48, 48)])
def test_batchnorm_train(C, N, H, W):
    x = (rnd(N, C, H, W, seed=1) * 2 + 3).requires_grad_(True)
    gamma = (rnd(C, seed=2) * 0.3 + 1).requires_grad_(True)
    beta = rnd(C, seed=3).requires_grad_(True)
    rm, rv = rnd(C, seed=4), rnd(C, seed=5).abs() + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    res = rnd(N, C, H, W, seed=6)
    ybn = F.batch_norm(x, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    out_ref = F.relu(ybn + res)
    dout = rnd(N, C, H, W, seed=7)
    dx_ref, dg_ref, db_ref = torch.autograd.grad(out_ref, (x, gamma, beta), dout, retain_graph=True)
    M = N * H * W
    xs = nhwc(x.detach())
    dv = lambda t: t.detach().clone().to(DEV)
    g_, b_, rm_, rv_ = dv(gamma), dv(beta), dv(rm), dv(rv)
    sc, sh, mu, rs = (torch.empty(C, device=DEV) for _ in range(4))
    ws = reduce_ws()
    _lib.check(L().dbn_bn_train_stats(xs.data_ptr(), M, C, g_.data_ptr(), b_.data_ptr(), 1e-5, 0.1, rm_.data_ptr(),
                                      rv_.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), ws.data_ptr(),
                                      stream()), 'bn stats')
    report('bn running_mean', rm_.cpu(), rm_ref, 1e-5, 1e-5)
    report('bn running_var', rv_.cpu(), rv_ref, 1e-5, 1e-5)
    ress = nhwc(res)
    out = torch.empty_like(xs)
    _lib.check(L().dbn_bn_apply(xs.data_ptr(), sc.data_ptr(), sh.data_ptr(), ress.data_ptr(), None, None, out.data_ptr(), M, C, 1,
                                stream()), 'bn apply')
    report('bn apply+res+relu', nchw(out), out_ref, 1e-5, 1e-5)
    douts = nhwc(dout)
    dy = torch.empty_like(xs)
    gout = torch.ones_like(xs)
    dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    _lib.check(L().dbn_bn_backward(xs.data_ptr(), out.data_ptr(), None, None, douts.data_ptr(), mu.data_ptr(), rs.data_ptr(),
                                   g_.data_ptr(), dy.data_ptr(), gout.data_ptr(), 1, dg.data_ptr(), db.data_ptr(), M, C, 1.0,
                                   ws.data_ptr(), stream()), 'bn bwd')
    report('bn bwd dx', nchw(dy), dx_ref, 2e-5, 1e-4)
    report('bn bwd dgamma', dg.cpu(), dg_ref, 1e-4, 1e-4)
    report('bn bwd dbeta', db.cpu(), db_ref, 1e-4, 1e-4)
    report('bn bwd masked grad (acc)', nchw(gout), dout * (out_ref > 0) + 1, 1e-6, 1e-6)
    # plain BN+ReLU (no residual): mask recomputed from y with the forward's scale/shift == mask from the saved activation
    zr_ref = F.relu(ybn)
    dxr_ref, dgr_ref, dbr_ref = torch.autograd.grad(zr_ref, (x, gamma, beta), dout, retain_graph=True)
    for use_saved in (True, False):
        zs = nhwc(zr_ref.detach())
        _lib.check(L().dbn_bn_backward(xs.data_ptr(), zs.data_ptr() if use_saved else None, None if use_saved else sc.data_ptr(),
                                       None if use_saved else sh.data_ptr(), douts.data_ptr(), mu.data_ptr(), rs.data_ptr(),
                                       g_.data_ptr(), dy.data_ptr(), None, 0, dg.data_ptr(), db.data_ptr(), M, C, 1.0, ws.data_ptr(),
                                       stream()), 'bn bwd relu')
        report('bn+relu bwd dx (saved mask=%s)' % use_saved, nchw(dy), dxr_ref, 2e-5, 1e-4)
        report('bn+relu bwd dgamma', dg.cpu(), dgr_ref, 1e-4, 1e-4)
        report('bn+relu bwd dbeta', db.cpu(), dbr_ref, 1e-4, 1e-4)
    # second-BN residual (downsample form) and eval coefficients
    sc2, sh2 = rnd(C, seed=8).to(DEV), rnd(C, seed=9).to(DEV)
    _lib.check(L().dbn_bn_apply(xs.data_ptr(), sc.data_ptr(), sh.data_ptr(), ress.data_ptr(), sc2.data_ptr(), sh2.data_ptr(),
                                out.data_ptr(), M, C, 0, stream()), 'bn apply2')
    ref2 = ybn + res * sc2.cpu().view(1, C, 1, 1) + sh2.cpu().view(1, C, 1, 1)
    report('bn apply + bn(res)', nchw(out), ref2, 1e-5, 1e-5)
    _lib.check(L().dbn_bn_eval_coef(C, g_.data_ptr(), b_.data_ptr(), rm_.data_ptr(), rv_.data_ptr(), 1e-5, sc.data_ptr(),
                                    sh.data_ptr(), stream()), 'bn eval')
    _lib.check(L().dbn_bn_apply(xs.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, None, None, out.data_ptr(), M, C, 0, stream()),
               'bn apply3')
    report('bn eval', nchw(out), F.batch_norm(x, rm_ref, rv_ref, gamma, beta, False, 0.1, 1e-5), 1e-5, 1e-5)
    cs = torch.empty(C, device=DEV)
    _lib.check(L().dbn_col_sum(xs.data_ptr(), M, C, cs.data_ptr(), 1.0, ws.data_ptr(), stream()), 'col_sum')
    report('col_sum', cs.cpu(), x.detach().sum((0, 2, 3)), 1e-5 * M, 1e-5)


@pytest.mark.parametrize('N,H,W', [(2, 16, 12), (1, 7, 9), (1, 32, 32)])
def test_bnrelu_maxpool(N, H, W):
    C = 64
    y = rnd(N, C, H, W, seed=1).requires_grad_(True)
    sc, sh = rnd(C, seed=2) * 0.5 + 1, rnd(C, seed=3) * 0.3
    z = F.relu(y * sc.view(1, C, 1, 1) + sh.view(1, C, 1, 1))
    pool_ref = F.max_pool2d(z, 3, 2, 1)
    dp = rnd(*pool_ref.shape, seed=4)
    (dz_ref, ) = torch.autograd.grad(pool_ref, z, dp)
    dz_ref = dz_ref * (z > 0)
    ys, scd, shd = nhwc(y.detach()), sc.to(DEV), sh.to(DEV)
    Ho, Wo = pool_ref.shape[2:]
    pool = torch.empty(N, Ho, Wo, C, device=DEV)
    _lib.check(L().dbn_bnrelu_maxpool_fwd(ys.data_ptr(), scd.data_ptr(), shd.data_ptr(), pool.data_ptr(), N, H, W, C, stream()),
               'pool')
    report('maxpool fwd', nchw(pool), pool_ref, 1e-6, 1e-6)
    dz = torch.empty(N, H, W, C, device=DEV)
    dps = nhwc(dp)
    _lib.check(L().dbn_bnrelu_maxpool_bwd(ys.data_ptr(), scd.data_ptr(), shd.data_ptr(), pool.data_ptr(), dps.data_ptr(),
                                          dz.data_ptr(), N, H, W, C, stream()), 'pool bwd')
    report('maxpool bwd', nchw(dz), dz_ref, 1e-6, 1e-6)


@pytest.mark.parametrize('hs,ws,h,w', [(4, 4, 8, 8), (2, 3, 8, 12), (1, 1, 8, 8), (3, 5, 7, 9), (8, 8, 8, 8)])
def test_nearest_upsample(hs, ws, h, w):
    N, C = 2, 64
    a = rnd(N, C, hs, ws, seed=1).requires_grad_(True)
    b = rnd(N, C, h, w, seed=2)
    ref = F.interpolate(a, size=(h, w)) + b
    dout = rnd(N, C, h, w, seed=3)
    (da_ref, ) = torch.autograd.grad(ref, a, dout)
    as_, bs = nhwc(a.detach()), nhwc(b)
    out = torch.empty(N, h, w, C, device=DEV)
    _lib.check(L().dbn_nearest_up_fwd(as_.data_ptr(), bs.data_ptr(), out.data_ptr(), N, hs, ws, C, h, w, C, 0, stream()), 'up')
    report('upsample_add fwd', nchw(out), ref, 1e-6, 1e-6)
    cat = torch.zeros(N, h, w, 256, device=DEV)
    _lib.check(L().dbn_nearest_up_fwd(as_.data_ptr(), None, cat.data_ptr(), N, hs, ws, C, h, w, 256, 128, stream()), 'upcat')
    report('upsample_cat fwd', nchw(cat)[:, 128:192], F.interpolate(a.detach(), size=(h, w)), 1e-6, 1e-6)
    assert float(cat[..., :128].abs().max()) == 0 and float(cat[..., 192:].abs().max()) == 0
    dbig = torch.zeros(N, h, w, 256, device=DEV)
    dbig[..., 64:128] = nhwc(dout)
    da = torch.ones(N, hs, ws, C, device=DEV)
    _lib.check(L().dbn_nearest_up_bwd(dbig.data_ptr(), da.data_ptr(), N, hs, ws, C, h, w, 256, 64, 1, stream()), 'up bwd')
    report('upsample bwd (acc)', nchw(da), da_ref + 1, 1e-5, 1e-5)


def test_input_pack():
    x = rnd(2, 3, 10, 12, seed=1)
    out = torch.empty(2, 10, 12, 4, device=DEV)
    xd = x.to(DEV)
    _lib.check(L().dbn_nchw3_to_nhwc4(xd.data_ptr(), out.data_ptr(), 2, 10, 12, stream()), 'pack input')
    ref = torch.cat([x, torch.zeros(2, 1, 10, 12)], 1)
    report('nchw3->nhwc4', nchw(out), ref, 0, 0)


@pytest.mark.parametrize('N,Hq,Wq,ch', [(2, 8, 6, 3), (1, 5, 7, 2), (3, 16, 16, 3)])
def test_head_tail(N, Hq, Wq, ch):
    xb = rnd(N, 64, Hq, Wq, seed=1).abs().requires_grad_(True)
    xt = rnd(N, 64, Hq, Wq, seed=2).abs().requires_grad_(True)
    wb = rnd(64, 1, 2, 2, seed=3, scale=0.2).requires_grad_(True)
    wt = rnd(64, 1, 2, 2, seed=4, scale=0.2).requires_grad_(True)
    bb, bt = torch.tensor([0.1], requires_grad=True), torch.tensor([-0.2], requires_grad=True)
    P = torch.sigmoid(F.conv_transpose2d(xb, wb, bb, 2))
    T = torch.sigmoid(F.conv_transpose2d(xt, wt, bt, 2))
    if ch == 3:
        ref = torch.cat([P, T, torch.reciprocal(1 + torch.exp(-50 * (P - T)))], 1)
    else:
        ref = torch.cat([P, T], 1)
    dpred = rnd(*ref.shape, seed=5)
    grads = torch.autograd.grad(ref, (xb, xt, wb, bb, wt, bt), dpred)
    d = lambda t: t.detach().contiguous().to(DEV)
    xbs, xts = nhwc(xb.detach()), nhwc(xt.detach())
    wbd, wtd, bbd, btd = d(wb), d(wt), d(bb), d(bt)
    out = torch.full(ref.shape, float('nan'), device=DEV)
    _lib.check(L().dbn_head_tail_fwd(xbs.data_ptr(), xts.data_ptr(), wbd.data_ptr(), wtd.data_ptr(), bbd.data_ptr(),
                                     btd.data_ptr(), None, None, None, None, out.data_ptr(), N, Hq, Wq, ch, 50.0, stream()), 'head fwd')
    report('head tail fwd', out.cpu(), ref, 2e-6, 1e-5)
    if ch != 3:
        return
    dxb, dxt = torch.empty_like(xbs), torch.empty_like(xts)
    dwb, dwt = torch.empty(256, device=DEV), torch.empty(256, device=DEV)
    dbb, dbt = torch.empty(1, device=DEV), torch.empty(1, device=DEV)
    ws = torch.empty(L().dbn_head_tail_bwd_ws_floats(), device=DEV)
    dpd = d(dpred)
    _lib.check(L().dbn_head_tail_bwd(xbs.data_ptr(), xts.data_ptr(), wbd.data_ptr(), wtd.data_ptr(), out.data_ptr(),
                                     dpd.data_ptr(), *([None] * 9), dxb.data_ptr(), dxt.data_ptr(), dwb.data_ptr(), dbb.data_ptr(),
                                     dwt.data_ptr(), dbt.data_ptr(), N, Hq, Wq, ch, 50.0, 1.0, ws.data_ptr(), stream()),
               'head bwd')
    sc = 1e-4
    report('head bwd dxb', nchw(dxb), grads[0], sc, 1e-3)
    report('head bwd dxt', nchw(dxt), grads[1], sc, 1e-3)
    report('head bwd dwb', dwb.cpu().view(64, 1, 2, 2), grads[2], sc * 10, 1e-3)
    report('head bwd dbias_b', dbb.cpu(), grads[3], sc * 10, 1e-3)
    report('head bwd dwt', dwt.cpu().view(64, 1, 2, 2), grads[4], sc * 10, 1e-3)
    report('head bwd dbias_t', dbt.cpu(), grads[5], sc * 10, 1e-3)


def test_adam():
    from oracle import dbnet_oracle as O
    n = 1003 * 4
    p0, g1, g2 = rnd(n, seed=1), rnd(n, seed=2) * 1e-3, rnd(n, seed=3) * 1e-2
    sd = {'w': p0.clone()}
    opt = O.AdamState(lr=0.005)
    opt.step(sd, {'w': g1})
    opt.step(sd, {'w': g2})
    p, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for i, g in enumerate((g1, g2)):
        gd = (g * 4).to(DEV)  # grad_scale 0.25 undoes the x4 (the 1/world path)
        _lib.check(L().dbn_adam_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 0.005, 0.9, 0.999, 1e-8, i + 1,
                                     0.25, stream()), 'adam')
    report('adam params', p.cpu(), sd['w'], 1e-6, 1e-5)
    report('adam exp_avg', m.cpu(), opt.m['w'], 1e-8, 1e-5)


# ---- split-bf16 math modes (same kernels, products on the bf16 matrix pipe) ---------------------
# ns=3 ("bf16x3"): fp32-accurate -> same tolerances as the native fp32 path.  ns=1: bf16 operands
# (8-bit mantissa): relative error ~ 2^-9 * sqrt(K)/sqrt(K) per output -> 2e-2 of the output scale.
SPLIT_TOL = {3: (1e-4, 1e-4), 1: (3e-2, 3e-2)}


@pytest.mark.parametrize('ns', [3, 1])
@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('tile', [0, 1, 2, 4])
def test_conv_forward_split(case, tile, ns):
    N, Ci, Co, k, s, p, H, W = case
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, k, k, seed=2, scale=(2.0 / (Ci * k * k))**0.5)
    b = rnd(Co, seed=3)
    ref = F.conv2d(x, w, b, s, p)
    Ho, Wo = ref.shape[2:]
    xs = nhwc(pad_c(x, (Ci + 3) // 4 * 4))
    y = torch.full((N, Ho, Wo, Co), float('nan'), device=DEV)
    igemm(xs, pack(w, 0, s, ns), b.to(DEV), y, k, s, p, 0, 0, tile, ns)
    atol, rtol = SPLIT_TOL[ns]
    report('conv fwd ns=%d %s tile %d' % (ns, case, tile), nchw(y), ref, atol * float(ref.abs().max()) if ns == 1 else atol, rtol)


@pytest.mark.parametrize('ns', [3, 1])
@pytest.mark.parametrize('case', DGRAD_CASES)
def test_conv_dgrad_split(case, ns):
    N, Ci, Co, k, s, p, H, W = case
    x = rnd(N, Ci, H, W, seed=1).requires_grad_(True)
    w = rnd(Co, Ci, k, k, seed=2, scale=(2.0 / (Ci * k * k))**0.5)
    y = F.conv2d(x, w, None, s, p)
    dy = rnd(*y.shape, seed=4)
    (dx_ref, ) = torch.autograd.grad(y, x, dy)
    dx = torch.full((N, H, W, Ci), float('nan'), device=DEV)
    igemm(nhwc(dy), pack(w, 1, s, ns), None, dx, k, s, p, 1, 0, 0, ns)
    atol, rtol = SPLIT_TOL[ns]
    report('conv dgrad ns=%d %s' % (ns, case), nchw(dx), dx_ref, atol * float(dx_ref.abs().max()) if ns == 1 else atol, rtol)


@pytest.mark.parametrize('ns', [3, 1])
@pytest.mark.parametrize('case', WGRAD_CASES)
def test_conv_wgrad_split(case, ns):
    N, Ci, Co, k, s, p, H, W = case
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, k, k, seed=2).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = rnd(*y.shape, seed=4)
    (dw_ref, ) = torch.autograd.grad(y, w, dy)
    g = wgrad(nhwc(dy), nhwc(pad_c(x, (Ci + 3) // 4 * 4)), Co, Ci, k, s, p, 1.0, ns)
    scale = float(dw_ref.abs().max())
    if ns == 3:
        report('conv wgrad ns=3 %s' % (case, ), g.cpu(), dw_ref, 2e-5 * scale + 1e-5, 1e-4)
    else:
        report('conv wgrad ns=1 %s' % (case, ), g.cpu(), dw_ref, 3e-2 * scale, 3e-2)


@pytest.mark.parametrize('ns', [3, 1])
def test_conv_transpose_split(ns):
    N, Ci, Co, H, W = 2, 64, 64, 8, 6
    x = rnd(N, Ci, H, W, seed=1).requires_grad_(True)
    w = rnd(Ci, Co, 2, 2, seed=2, scale=0.1).requires_grad_(True)
    b = rnd(Co, seed=3)
    ref = F.conv_transpose2d(x, w, b, 2)
    dy = rnd(*ref.shape, seed=5)
    dx_ref, dw_ref = torch.autograd.grad(ref, (x, w), dy)
    xs = nhwc(x.detach())
    atol, rtol = SPLIT_TOL[ns]
    y = torch.full((N, 2 * H, 2 * W, Co), float('nan'), device=DEV)
    igemm(xs, pack(w.detach(), 1, 2, ns), b.to(DEV), y, 2, 2, 0, 1, 0, 0, ns)
    report('convT fwd ns=%d' % ns, nchw(y), ref, atol * float(ref.abs().max()), rtol)
    dys = nhwc(dy)
    dx = torch.full((N, H, W, Ci), float('nan'), device=DEV)
    igemm(dys, pack(w.detach(), 0, 2, ns), None, dx, 2, 2, 0, 0, 0, 0, ns)
    report('convT dgrad ns=%d' % ns, nchw(dx), dx_ref, atol * float(dx_ref.abs().max()), rtol)
    g = wgrad(xs, dys, Ci, Co, 2, 2, 0, 1.0, ns)
    report('convT wgrad ns=%d' % ns, g.cpu(), dw_ref, atol * float(dw_ref.abs().max()), rtol)


@pytest.mark.parametrize('hs,ws,h,w', [(428, 640, 427, 640), (12, 20, 9, 17), (8, 8, 8, 8), (6, 5, 13, 11), (4, 4, 1, 1)])
def test_bilinear_align_corners(hs, ws, h, w):
    """models.py:43-46: F.interpolate(size=(H,W), mode='bilinear', align_corners=True) on NCHW planes, and its adjoint."""
    N, C = 2, 3
    a = rnd(N, C, hs, ws, seed=1).requires_grad_(True)
    ref = F.interpolate(a, size=(h, w), mode='bilinear', align_corners=True)
    dout = rnd(N, C, h, w, seed=2)
    (da_ref, ) = torch.autograd.grad(ref, a, dout)
    ad = a.detach().to(DEV)
    out = torch.full((N, C, h, w), float('nan'), device=DEV)
    _lib.check(L().dbn_bilinear_fwd(ad.data_ptr(), out.data_ptr(), N * C, hs, ws, h, w, stream()), 'bilinear fwd')
    report('bilinear fwd', out.cpu(), ref, 1e-5, 1e-5)
    dd = dout.to(DEV)
    da = torch.full((N, C, hs, ws), float('nan'), device=DEV)
    _lib.check(L().dbn_bilinear_bwd(dd.data_ptr(), da.data_ptr(), N * C, hs, ws, h, w, stream()), 'bilinear bwd')
    report('bilinear bwd', da.cpu(), da_ref, 1e-5, 1e-5)


@pytest.mark.parametrize('case', [(2, 64, 64, 3, 1, 1, 16, 12), (3, 128, 64, 1, 1, 0, 9, 7), (2, 3, 64, 7, 2, 3, 32, 40), (1, 256, 256, 3, 1, 1, 12, 12),
                                  (4, 64, 64, 3, 1, 1, 40, 40)])
@pytest.mark.parametrize('tile', [0, 1, 2, 4])
@pytest.mark.parametrize('ns', [0, 3])
def test_conv_with_fused_bn_statistics(case, tile, ns):
    """dbn_conv_bn_f32: conv output + train-mode BN coefficients/running stats from the epilogue partials == F.conv2d + F.batch_norm."""
    N, Ci, Co, k, s, p, H, W = case
    x = rnd(N, Ci, H, W, seed=1) * 2 + 0.5
    w = rnd(Co, Ci, k, k, seed=2, scale=(2.0 / (Ci * k * k))**0.5)
    b = rnd(Co, seed=3) * 3  # large bias: mean >> std exercises the pivot
    gamma, beta = rnd(Co, seed=4) * 0.3 + 1, rnd(Co, seed=5)
    rm, rv = rnd(Co, seed=6), rnd(Co, seed=7).abs() + 0.5
    y_ref = F.conv2d(x, w, b, s, p)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    z_ref = F.batch_norm(y_ref, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    Ho, Wo = y_ref.shape[2:]
    xs = nhwc(pad_c(x, (Ci + 3) // 4 * 4))
    y = torch.full((N, Ho, Wo, Co), float('nan'), device=DEV)
    d = lambda t: t.clone().to(DEV)
    g_, b_, rm_, rv_, bias_ = d(gamma), d(beta), d(rm), d(rv), d(b)
    sc, sh, mu, rs = (torch.empty(Co, device=DEV) for _ in range(4))
    ws = torch.empty(L().dbn_conv_bn_ws_floats(N, Ho, Wo, Co, 0, s), device=DEV)
    wpk = pack(w, 0, s, ns)
    _lib.check(L().dbn_conv_bn_f32(xs.data_ptr(), wpk.data_ptr(), bias_.data_ptr(), y.data_ptr(), N, H, W, xs.shape[3], Ho, Wo, Co, k, k,
                                   s, p, 0, 0, tile, ns, g_.data_ptr(), b_.data_ptr(), 1e-5, 0.1, rm_.data_ptr(), rv_.data_ptr(),
                                   sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), ws.data_ptr(), stream()), 'conv_bn')
    report('conv_bn y', nchw(y), y_ref, 1e-4, 1e-4)
    report('conv_bn running_mean', rm_.cpu(), rm_ref, 1e-5, 1e-5)
    report('conv_bn running_var', rv_.cpu(), rv_ref, 1e-5, 2e-5)
    z = nchw(y) * sc.cpu().view(1, Co, 1, 1) + sh.cpu().view(1, Co, 1, 1)
    report('conv_bn normalised output', z, z_ref, 2e-5, 1e-4)


def test_fpn_structured_conv_gradients():
    """Data/weight gradients of conv3x3 over [p2 | up2(p3) | up4(p4) | up8(p5)] via the combined (f+2)x(f+2) stride-f
    convs == autograd of F.conv2d(torch.cat(nearest-upsampled)) (segmentation_body.py:75-76,82-87)."""
    N, H, W, Cg, Co = 2, 16, 24, 64, 256
    ps = [rnd(N, Cg, H >> g, W >> g, seed=10 + g).requires_grad_(True) for g in range(4)]
    w = rnd(Co, 4 * Cg, 3, 3, seed=3, scale=0.03).requires_grad_(True)
    cat = torch.cat([ps[0]] + [F.interpolate(ps[g], size=(H, W)) for g in range(1, 4)], 1)
    y = F.conv2d(cat, w, None, 1, 1)
    dy = rnd(*y.shape, seed=4)
    grads = torch.autograd.grad(y, ps + [w], dy)
    dys = nhwc(dy)
    wd_ = w.detach().to(DEV)
    ts = []
    for g in range(4):
        f, k = 1 << g, (1 << g) + 2
        wdg = torch.empty(Cg, Co, k, k, device=DEV)
        _lib.check(L().dbn_fpn_combine_weights(wd_.data_ptr(), Co, 4 * Cg, g, Cg, wdg.data_ptr(), stream()), 'combine')
        d = torch.full((N, H >> g, W >> g, Cg), float('nan'), device=DEV)
        igemm(dys, pack(wdg.cpu(), 0, f), None, d, k, f, 1, 0)
        report('fpn level %d dgrad' % g, nchw(d), grads[g], 1e-4, 1e-4)
        ts.append(wgrad(nhwc(ps[g].detach()), dys, Cg, Co, k, f, 1))
    dw = torch.full((Co, 4 * Cg, 3, 3), float('nan'), device=DEV)
    _lib.check(L().dbn_fpn_scatter_wgrad(ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), Co, Cg, dw.data_ptr(),
                                         stream()), 'scatter')
    report('fpn structured wgrad', dw.cpu(), grads[4], 2e-5 * float(grads[4].abs().max()) + 1e-5, 1e-4)


@pytest.mark.parametrize('f,k,pad', [(2, 4, 1), (4, 6, 1), (8, 10, 1), (4, 4, 0), (2, 3, 1), (4, 3, 1), (8, 2, 0)])
@pytest.mark.parametrize('ns', [0, 3])
def test_conv_transpose_general_stride(f, k, pad, ns):
    """mode 1 with stride f in {2,4,8}: f*f output-parity classes in one launch == F.conv_transpose2d; classes without
    taps (k < f) are zero-filled; accumulate=1 adds onto dst."""
    N, Ci, Co, H, W = 2, 64, 128, 5, 7
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Ci, Co, k, k, seed=2, scale=0.1)
    b = rnd(Co, seed=3) if k >= f else None  # kernels smaller than the stride leave pixels without taps: no bias there
    ref = F.conv_transpose2d(x, w, b, f, pad)
    xs = nhwc(x)
    y = torch.full((N, ref.shape[2], ref.shape[3], Co), float('nan'), device=DEV)
    wpk = pack(w, 1, f, ns)
    igemm(xs, wpk, None if b is None else b.to(DEV), y, k, f, pad, 1, ns=ns)
    report('convT s%d k%d' % (f, k), nchw(y), ref, 1e-4, 1e-4)
    base = rnd(*ref.shape, seed=9)
    y2 = nhwc(base)
    igemm(xs, wpk, None, y2, k, f, pad, 1, accumulate=1, ns=ns)
    report('convT s%d k%d accumulate' % (f, k), nchw(y2), base + F.conv_transpose2d(x, w, None, f, pad), 1e-4, 1e-4)


@pytest.mark.parametrize('ns', [0, 3])
def test_fpn_structured_conv_forward(ns):
    """conv3x3 over [p2 | up2(p3) | up4(p4) | up8(p5)] == conv3x3(p2, W[:, :64]) + sum_g convT_{k=f+2, stride f, pad 1}(p_g, Wd_g),
    with the train-mode BN statistics of the accumulated result from the last launch (segmentation_body.py:75-76,82-87)."""
    N, H, W, Cg, Co = 2, 16, 24, 64, 256
    ps = [rnd(N, Cg, H >> g, W >> g, seed=10 + g) for g in range(4)]
    w = rnd(Co, 4 * Cg, 3, 3, seed=3, scale=0.03)
    bias = rnd(Co, seed=5)
    gamma, beta = rnd(Co, seed=6) * 0.3 + 1, rnd(Co, seed=7)
    rm, rv = rnd(Co, seed=8), rnd(Co, seed=9).abs() + 0.5
    cat = torch.cat([ps[0]] + [F.interpolate(ps[g], size=(H, W)) for g in range(1, 4)], 1)
    y_ref = F.conv2d(cat, w, bias, 1, 1)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    z_ref = F.batch_norm(y_ref, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    wd_ = w.to(DEV)
    y = torch.full((N, H, W, Co), float('nan'), device=DEV)
    igemm(nhwc(ps[0]), pack(w[:, :Cg].contiguous(), 0, 1, ns), bias.to(DEV), y, 3, 1, 1, 0, ns=ns)
    d = lambda t: t.clone().to(DEV)
    g_, b_, rm_, rv_ = d(gamma), d(beta), d(rm), d(rv)
    sc, sh, mu, rs = (torch.empty(Co, device=DEV) for _ in range(4))
    for g in range(1, 4):
        f, k = 1 << g, (1 << g) + 2
        wdg = torch.empty(Cg, Co, k, k, device=DEV)
        _lib.check(L().dbn_fpn_combine_weights(wd_.data_ptr(), Co, 4 * Cg, g, Cg, wdg.data_ptr(), stream()), 'combine')
        wpk = pack(wdg.cpu(), 1, f, ns)
        xs = nhwc(ps[g])
        if g < 3:
            igemm(xs, wpk, None, y, k, f, 1, 1, accumulate=1, ns=ns)
        else:
            ws = torch.empty(L().dbn_conv_bn_ws_floats(N, H, W, Co, 1, f), device=DEV)
            _lib.check(L().dbn_conv_bn_f32(xs.data_ptr(), wpk.data_ptr(), None, y.data_ptr(), N, H >> g, W >> g, Cg, H, W, Co, k, k, f, 1,
                                           1, 1, 0, ns, g_.data_ptr(), b_.data_ptr(), 1e-5, 0.1, rm_.data_ptr(), rv_.data_ptr(),
                                           sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), ws.data_ptr(), stream()), 'conv_bn')
    report('fpn fwd y', nchw(y), y_ref, 1e-4, 1e-4)
    report('fpn fwd running_mean', rm_.cpu(), rm_ref, 1e-5, 1e-5)
    report('fpn fwd running_var', rv_.cpu(), rv_ref, 1e-5, 2e-5)
    z = nchw(y) * sc.cpu().view(1, Co, 1, 1) + sh.cpu().view(1, Co, 1, 1)
    report('fpn fwd normalised output', z, z_ref, 2e-5, 1e-4)


@pytest.mark.parametrize('ns', [0, 3])
@pytest.mark.parametrize('shape', [(2, 16, 24, 64, 256), (1, 8, 8, 64, 128), (3, 40, 8, 32, 128)])
@pytest.mark.parametrize('bn', [False, True])
def test_pyramid_conv(shape, ns, bn):
    """dbn_pyramid_conv_f32 (one launch, no concat) == F.conv2d(torch.cat([p2, up2(p3), up4(p4), up8(p5)]), W, b, 1, 1)
    (+ train-mode BatchNorm coefficients / running statistics) — segmentation_body.py:75-76,82-87."""
    N, H, W, Cg, Co = shape
    ps = [rnd(N, Cg, H >> g, W >> g, seed=10 + g) for g in range(4)]
    w = rnd(Co, 4 * Cg, 3, 3, seed=3, scale=0.03)
    bias = rnd(Co, seed=5)
    gamma, beta = rnd(Co, seed=6) * 0.3 + 1, rnd(Co, seed=7)
    rm, rv = rnd(Co, seed=8), rnd(Co, seed=9).abs() + 0.5
    cat = torch.cat([ps[0]] + [F.interpolate(ps[g], size=(H, W)) for g in range(1, 4)], 1)
    y_ref = F.conv2d(cat, w, bias, 1, 1)
    wd_ = w.to(DEV)
    wpk = []
    for g in range(4):
        k = (1 << g) + 2
        wdg = torch.empty(Cg, Co, k, k, device=DEV)
        _lib.check(L().dbn_fpn_combine_weights(wd_.data_ptr(), Co, 4 * Cg, g, Cg, wdg.data_ptr(), stream()), 'combine')
        wpk.append(pack(wdg.cpu(), 1, 1 << g, ns))
    xs = [nhwc(t) for t in ps]
    y = torch.full((N, H, W, Co), float('nan'), device=DEV)
    d = lambda t: t.clone().to(DEV)
    g_, b_, rm_, rv_, bias_ = d(gamma), d(beta), d(rm), d(rv), d(bias)
    sc, sh, mu, rs = (torch.empty(Co, device=DEV) for _ in range(4))
    ws = torch.empty(L().dbn_pyramid_conv_ws_floats(N, H, W, Co), device=DEV)
    bnargs = (g_.data_ptr(), b_.data_ptr(), 1e-5, 0.1, rm_.data_ptr(), rv_.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(),
              rs.data_ptr(), ws.data_ptr()) if bn else (None, None, 0.0, 0.0) + (None, ) * 7
    _lib.check(L().dbn_pyramid_conv_f32(*[t.data_ptr() for t in xs], *[t.data_ptr() for t in wpk], bias_.data_ptr(), y.data_ptr(), N, H,
                                        W, Cg, Co, 0, ns, *bnargs, stream()), 'pyramid_conv')
    report('pyramid conv y', nchw(y), y_ref, 1e-4, 1e-4)
    if bn:
        rm_ref, rv_ref = rm.clone(), rv.clone()
        z_ref = F.batch_norm(y_ref, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
        report('pyramid running_mean', rm_.cpu(), rm_ref, 1e-5, 1e-5)
        report('pyramid running_var', rv_.cpu(), rv_ref, 1e-5, 2e-5)
        z = nchw(y) * sc.cpu().view(1, Co, 1, 1) + sh.cpu().view(1, Co, 1, 1)
        report('pyramid normalised output', z, z_ref, 2e-5, 1e-4)


@pytest.mark.parametrize('case', [(2, 256, 64, 10, 8, 1, 40, 40, 0), (2, 256, 64, 6, 4, 1, 24, 40, 0), (1, 512, 512, 3, 1, 1, 10, 12, 0),
                                  (2, 128, 64, 3, 1, 1, 9, 7, 1), (1, 64, 64, 3, 2, 1, 17, 15, 0)])
@pytest.mark.parametrize('ksplit', [2, 5, 7])
@pytest.mark.parametrize('ns', [0, 3])
def test_conv_splitk(case, ksplit, ns):
    """dbn_igemm_splitk_f32 (reduction split over workgroup rows + fixed-order slab sum, bias, accumulate) == F.conv2d /
    its data gradient; equals the unsplit kernel to rounding."""
    N, Ci, Co, k, s, p, H, W, mode = case
    x = rnd(N, Ci, H, W, seed=1).requires_grad_(mode == 1)
    w = rnd(Co, Ci, k, k, seed=2, scale=(1.0 / (Ci * k * k))**0.5)
    b = rnd(Co, seed=3)
    if mode == 0:
        ref = F.conv2d(x, w, b, s, p)
        src, wpk, bias, Cd = nhwc(x.detach()), pack(w, 0, s, ns), b.to(DEV), Co
    else:
        y = F.conv2d(x, w, None, s, p)
        dy = rnd(*y.shape, seed=4)
        (ref, ) = torch.autograd.grad(y, x, dy)
        src, wpk, bias, Cd = nhwc(dy), pack(w, 1, s, ns), None, Ci
    base = rnd(*ref.shape, seed=9)
    dst = nhwc(base)
    Nn, Hs, Ws, Cs = src.shape
    _, Hd, Wd, _ = dst.shape
    slab = torch.full((L().dbn_igemm_splitk_slab_floats(ksplit, dst.shape[0], dst.shape[1], dst.shape[2], dst.shape[3]), ), float('nan'), device=DEV)
    _lib.check(L().dbn_igemm_splitk_f32(src.data_ptr(), wpk.data_ptr(), None if bias is None else bias.data_ptr(), dst.data_ptr(), Nn, Hs,
                                        Ws, Cs, Hd, Wd, Cd, k, k, s, p, mode, 1, 0, ns, ksplit, slab.data_ptr(), stream()), 'splitk')
    report('splitk conv %s ks=%d' % (case, ksplit), nchw(dst), base + ref.detach(), 1e-4, 1e-4)


def test_splitk_plan():
    plan = L().dbn_igemm_splitk_plan
    assert plan(16 * 20 * 20, 64, 100 * 256, 256) > 1  # FPN level-3 data gradient: 100 tiles, K = 25600
    assert plan(16 * 160 * 160, 256, 2304, 256) == 1  # plenty of tiles
    assert plan(16 * 20 * 20, 64, 100 * 4, 4) == 1  # K order of thin inputs is not splittable


@pytest.mark.parametrize('case', [(2, 64, 64, 9, 11, 1), (1, 128, 64, 12, 10, 2), (2, 64, 128, 7, 7, 1)])
@pytest.mark.parametrize('zero_offsets', [False, True])
def test_deformable_conv(case, zero_offsets):
    """conv2_offset -> DeformConv2d (resnet.py:61-65,81-82,119-124,145-146) as deformable im2col + 1x1 GEMM, and its three
    gradients, vs the restated DCNv1 op (oracle.deform_conv2d + autograd).  Zero offsets (the reference's initial
    state, resnet.py:204-208) must reproduce the plain conv."""
    from oracle import dbnet_oracle as O
    N, C, Co, H, W, stride = case
    x = rnd(N, C, H, W, seed=1).requires_grad_(True)
    w = rnd(Co, C, 3, 3, seed=2, scale=(2.0 / (9 * C))**0.5).requires_grad_(True)
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    off = (torch.zeros(N, 18, Ho, Wo) if zero_offsets else rnd(N, 18, Ho, Wo, seed=3) * 1.5).requires_grad_(True)
    ref = O.deform_conv2d(x, off, w, stride, 1)
    if zero_offsets:
        report('zero offsets == conv2d', ref.detach(), F.conv2d(x, w, None, stride, 1).detach(), 1e-5, 1e-5)
    dy = rnd(*ref.shape, seed=4)
    dx_ref, doff_ref, dw_ref = torch.autograd.grad(ref, (x, off, w), dy)
    OS = 64  # offsets live in the 64-channel output of the zero-padded offset conv
    xs = nhwc(x.detach())
    offs = torch.zeros(N, Ho, Wo, OS, device=DEV)
    offs[..., :18] = nhwc(off.detach())
    cols = torch.full((N, Ho, Wo, 9 * C), float('nan'), device=DEV)
    dims = (N, H, W, C, Ho, Wo, 3, 3, stride, 1, OS)
    _lib.check(L().dbn_deform_im2col(xs.data_ptr(), offs.data_ptr(), cols.data_ptr(), *dims, stream()), 'deform_im2col')
    wp = torch.empty(Co, 9 * C, 1, 1, device=DEV)
    wd_ = w.detach().to(DEV)
    _lib.check(L().dbn_permute_weight(wd_.data_ptr(), wp.data_ptr(), Co, C, 9, 1, 1.0, stream()), 'permute')
    y = torch.full((N, Ho, Wo, Co), float('nan'), device=DEV)
    igemm(cols, pack(wp.cpu(), 0), None, y, 1, 1, 0, 0)
    report('deform conv fwd', nchw(y), ref.detach(), 1e-4, 1e-4)
    dys = nhwc(dy)
    dcols = torch.full((N, Ho, Wo, 9 * C), float('nan'), device=DEV)
    igemm(dys, pack(wp.cpu(), 1), None, dcols, 1, 1, 0, 1)
    ws = torch.empty(L().dbn_deform_col2im_ws_bytes(N, H, W, C, Ho, Wo, 3, 3), device=DEV, dtype=torch.uint8)
    runs = []
    for rep in range(3):  # the adjoint accumulates in 64-bit fixed point: bit-identical from run to run, whatever the offsets
        dx = torch.full((N, H, W, C), float('nan'), device=DEV)
        doffs = torch.full((N, Ho, Wo, OS), float('nan'), device=DEV)
        _lib.check(L().dbn_deform_col2im(dcols.data_ptr(), xs.data_ptr(), offs.data_ptr(), dx.data_ptr(), doffs.data_ptr(), 0,
                                         ws.data_ptr(), *dims, stream()), 'deform_col2im')
        runs.append((dx.clone(), doffs.clone()))
    assert all(torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]) for r in runs[1:]), 'col2im is not bit-reproducible'
    dx1 = torch.ones(N, H, W, C, device=DEV)  # accumulate form
    _lib.check(L().dbn_deform_col2im(dcols.data_ptr(), xs.data_ptr(), offs.data_ptr(), dx1.data_ptr(), doffs.data_ptr(), 1,
                                     ws.data_ptr(), *dims, stream()), 'deform_col2im acc')
    report('deform conv dx (accumulate)', nchw(dx1), dx_ref + 1, 1e-4, 1e-4)
    report('deform conv dx', nchw(dx), dx_ref, 1e-4, 1e-4)
    report('deform conv doffset', nchw(doffs[..., :18].contiguous()), doff_ref, 2e-4 * float(doff_ref.abs().max()) + 1e-5, 1e-4)
    assert float(doffs[..., 18:].abs().max()) == 0.0
    gp = wgrad(dys, cols, Co, 9 * C, 1, 1, 0)  # [Co, 9C, 1, 1] in (tap, channel) column order
    g = torch.empty(Co, C, 3, 3, device=DEV)
    _lib.check(L().dbn_permute_weight(gp.data_ptr(), g.data_ptr(), Co, C, 9, 0, 1.0, stream()), 'permute back')
    report('deform conv dweight', g.cpu(), dw_ref, 2e-5 * float(dw_ref.abs().max()) + 1e-5, 1e-4)


@pytest.mark.parametrize('case', [(2, 64, 9, 11, 1, 1.5), (1, 128, 12, 10, 2, 1.5), (2, 128, 17, 19, 1, 0.0), (1, 256, 13, 9, 1, 0.4), (1, 512, 8, 8, 2, 3.7),
                                  (2, 64, 21, 6, 1, 6.0), (1, 192, 10, 10, 1, 0.9)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_deformable_col2im_as_a_gather(case, dtype):
    """dbn_deform_col2im_gather_t (round 5): the adjoint of DeformConv2d's sampling (resnet.py:61-65,119-124) as a per-pixel gather in plain
    fp32 — vs autograd of the restated op (oracle.deform_conv2d) on the operands as stored, vs round 3's fixed-point scatter
    (dbn_deform_col2im_t: exact sums) at fp32 rounding of the sums, bit-identical from run to run, accumulate form, zero padding channels.
    Offsets up to 6 pixels (window search bound taken on the device), stride 1 / 2, odd maps, C = 64 ... 512."""
    from oracle import dbnet_oracle as O
    N, C, H, W, stride, oscale = case
    at = AT_OF[dtype]
    rq = lambda t: t.to(dtype).float()
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    x = rq(rnd(N, C, H, W, seed=1)).requires_grad_(True)
    off = rq(rnd(N, 18, Ho, Wo, seed=3) * oscale).requires_grad_(True)
    dcols_ = rq(rnd(N, Ho, Wo, 9 * C, seed=4))
    # cols = sampled columns [N, Ho, Wo, 9, C]: the 1x1 GEMM's input — the adjoint of the sampling is autograd of <cols, dcols>
    eye = torch.zeros(9 * C, C, 3, 3)
    for k in range(9):
        eye[k * C + torch.arange(C), torch.arange(C), k // 3, k % 3] = 1.0
    cols_ref = O.deform_conv2d(x.double(), off.double(), eye.double(), stride, 1)  # [N, 9C, Ho, Wo]
    dx_ref, doff_ref = torch.autograd.grad(cols_ref, (x, off), dcols_.permute(0, 3, 1, 2).double())
    OS = 64
    xs = nhwc(x.detach()).to(dtype)
    offs = torch.zeros(N, Ho, Wo, OS, device=DEV, dtype=dtype)
    offs[..., :18] = nhwc(off.detach()).to(dtype)
    dcols = dcols_.to(DEV).to(dtype)
    dims = (N, H, W, C, Ho, Wo, 3, 3, stride, 1, OS)
    ws = torch.empty(L().dbn_deform_col2im_gather_ws_bytes(N, Ho, Wo), device=DEV, dtype=torch.uint8).fill_(0xff)  # (needs no initialisation)
    runs = []
    for rep in range(3):
        dx = torch.full((N, H, W, C), float('nan'), device=DEV, dtype=dtype)
        doffs = torch.full((N, Ho, Wo, OS), float('nan'), device=DEV, dtype=dtype)
        _lib.check(L().dbn_deform_col2im_gather_t(at, dcols.data_ptr(), xs.data_ptr(), offs.data_ptr(), dx.data_ptr(), doffs.data_ptr(), 0,
                                                  ws.data_ptr(), *dims, stream()), 'deform_col2im_gather')
        runs.append((dx.clone(), doffs.clone()))
    assert all(torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]) for r in runs[1:]), 'the gather is not bit-reproducible'
    eps = 2.0**-8 if dtype == torch.bfloat16 else 2.0**-20  # (one rounding to the storage type | fp32 positions, weights and sums against fp64)
    sdx, sdo = float(dx_ref.abs().max()), float(doff_ref.abs().max())
    report('gather dx %s' % (case, ), nchw(dx.float()), dx_ref, eps * sdx, eps)
    report('gather doffset %s' % (case, ), nchw(doffs[..., :18].float().contiguous()), doff_ref, eps * sdo * 4 + 1e-6, eps)
    assert float(doffs[..., 18:].float().abs().max()) == 0.0
    # round 3's scatter (exact fixed-point sums, one rounding): the same numbers up to the fp32 rounding of the gather's sums
    ws3 = torch.empty(L().dbn_deform_col2im_ws_bytes(N, H, W, C, Ho, Wo, 3, 3), device=DEV, dtype=torch.uint8)
    dx3 = torch.full((N, H, W, C), float('nan'), device=DEV, dtype=dtype)
    do3 = torch.full((N, Ho, Wo, OS), float('nan'), device=DEV, dtype=dtype)
    _lib.check(L().dbn_deform_col2im_t(at, dcols.data_ptr(), xs.data_ptr(), offs.data_ptr(), dx3.data_ptr(), do3.data_ptr(), 0, ws3.data_ptr(),
                                       *dims, stream()), 'deform_col2im')
    report('gather vs scatter dx', dx.float().cpu(), dx3.float().cpu(), 2 * eps * sdx, 2 * eps)
    report('gather vs scatter doffset', doffs.float().cpu(), do3.float().cpu(), 8 * eps * sdo + 1e-6, 2 * eps)
    base = rq(rnd(N, H, W, C, seed=9)).to(DEV).to(dtype)  # accumulate form: dx += ...
    dx1 = base.clone()
    _lib.check(L().dbn_deform_col2im_gather_t(at, dcols.data_ptr(), xs.data_ptr(), offs.data_ptr(), dx1.data_ptr(), doffs.data_ptr(), 1,
                                              ws.data_ptr(), *dims, stream()), 'deform_col2im_gather acc')
    report('gather dx (accumulate)', nchw(dx1.float()), dx_ref + nchw(base.float()).double(), eps * (sdx + 3), eps)


def test_deformable_col2im_gather_non_finite_values_stay_local():
    """A NaN in dcols reaches the dx / doffset elements its sample touches and nothing else; a NaN offset drops its sample (outside by the
    comparison rules, as in the forward) and leaves every other element as it was."""
    N, C, H, W = 1, 64, 12, 12
    xs = nhwc(rnd(N, C, H, W, seed=1))
    offs = torch.zeros(N, H, W, 64, device=DEV)
    offs[..., :18] = (rnd(N, H, W, 18, seed=2) * 0.8).to(DEV)
    dcols = rnd(N, H, W, 9 * C, seed=3).to(DEV)
    dims = (N, H, W, C, H, W, 3, 3, 1, 1, 64)
    ws = torch.zeros(L().dbn_deform_col2im_gather_ws_bytes(N, H, W), device=DEV, dtype=torch.uint8)

    def run(dc, of):
        dx = torch.full((N, H, W, C), float('nan'), device=DEV)
        do = torch.full((N, H, W, 64), float('nan'), device=DEV)
        _lib.check(L().dbn_deform_col2im_gather_t(0, dc.data_ptr(), xs.data_ptr(), of.data_ptr(), dx.data_ptr(), do.data_ptr(), 0, ws.data_ptr(),
                                                  *dims, stream()), 'gather')
        return dx.cpu(), do.cpu()

    dx0, do0 = run(dcols, offs)
    assert torch.isfinite(dx0).all() and torch.isfinite(do0).all()
    d2 = dcols.clone()
    d2[0, 5, 6, 4 * C + 7] = float('nan')  # sample (pixel (5, 6), tap 4), channel 7
    dx, do = run(d2, offs)
    bad = torch.isnan(dx)
    assert 1 <= int(bad.sum()) <= 4 and bool(bad[0, 4:8, 5:9, 7].any()) and int(bad[..., 7].sum()) == int(bad.sum())
    assert torch.equal(dx[~bad], dx0[~bad])
    assert torch.isnan(do[0, 5, 6, 8:10]).all() and int(torch.isnan(do).sum()) == 2
    o2 = offs.clone()
    o2[0, 3, 3, 2] = float('nan')  # dy of tap 1 at pixel (3, 3)
    dx, do = run(dcols, o2)
    assert torch.isfinite(dx).all()
    assert float((dx - dx0).abs().max()) > 0 and int(((dx - dx0).abs() > 0).sum()) <= 4 * C  # that sample's contributions are gone, nothing else moved
    assert float(do[0, 3, 3, 2:4].abs().max()) == 0.0


def test_deformable_col2im_non_finite_and_outliers():
    """The sampling adjoint accumulates in 64-bit fixed point (deterministic), which by itself would turn a NaN / Inf column gradient
    into finite garbage (to_fixed(NaN) = 0): a non-finite element of dcols must come out as NaN in ALL of dx and doffset, a
    non-finite x as NaN in doffset — a diverged step stays visible (the float atomics of rounds 1-2 propagated it).  And the stated
    resolution: with one 1e6 outlier in dcols every other element keeps an absolute error <= 9 taps * 4 corners * 2^-44 * 2^20 ~ 2e-6
    (advisor finding, round 3; resnet.py:111-124 is the layer)."""
    N, C, H, W = 1, 64, 10, 12
    x = rnd(N, C, H, W, seed=1)
    off = rnd(N, 18, H, W, seed=3) * 1.5
    OS = 64
    xs = nhwc(x.detach())
    offs = torch.zeros(N, H, W, OS, device=DEV)
    offs[..., :18] = nhwc(off.detach())
    dims = (N, H, W, C, H, W, 3, 3, 1, 1, OS)
    ws = torch.empty(L().dbn_deform_col2im_ws_bytes(N, H, W, C, H, W, 3, 3), device=DEV, dtype=torch.uint8)

    def col2im(dcols, xin=xs):
        dx = torch.zeros(N, H, W, C, device=DEV)
        doffs = torch.zeros(N, H, W, OS, device=DEV)
        _lib.check(L().dbn_deform_col2im(dcols.data_ptr(), xin.data_ptr(), offs.data_ptr(), dx.data_ptr(), doffs.data_ptr(), 0, ws.data_ptr(),
                                         *dims, stream()), 'deform_col2im')
        torch.cuda.synchronize()
        return dx, doffs[..., :18]

    dcols = nhwc(rnd(N, 9 * C, H, W, seed=5)).contiguous()
    dx0, do0 = col2im(dcols)
    assert torch.isfinite(dx0).all() and torch.isfinite(do0).all()
    for bad in (float('nan'), float('inf')):
        d2 = dcols.clone()
        d2[0, 3, 4, 17] = bad
        dx, do = col2im(d2)
        assert torch.isnan(dx).all() and torch.isnan(do).all(), 'a non-finite column gradient must not come out finite'
    x2 = xs.clone()
    x2[0, 2, 2, 5] = float('nan')
    dx, do = col2im(dcols, x2)
    assert torch.isnan(do).all() and torch.equal(dx, dx0)  # dx does not depend on x
    # one huge outlier: the reference adjoint (autograd through the restated DCNv1 sampling) on the same column gradients
    d3 = dcols.clone()
    d3[0, 5, 6, 100] = 1.0e6
    dx, do = col2im(d3)
    dxa, _ = col2im(dcols)
    diff = (dx - dxa).abs()
    touched = diff > 1e-3  # the few input pixels the outlier's sample reaches (<= 4 corners x 1 channel)
    assert 1 <= int(touched.sum()) <= 4
    assert float(diff[~touched].max()) <= 4e-6, float(diff[~touched].max())  # everything else: the stated resolution floor


def test_head_tail_with_fused_batchnorm_relu():
    """bn_scale/shift given: the kernels take the PRE-BN ConvTranspose outputs and apply BN + ReLU on load
    (segmentation_head.py:27-29,74-79); gradients are w.r.t. the post-ReLU activations."""
    N, Hq, Wq = 2, 9, 7
    yb, yt = rnd(N, 64, Hq, Wq, seed=1), rnd(N, 64, Hq, Wq, seed=2)
    sc = [rnd(64, seed=10 + i) * 0.3 + 1 for i in range(2)]
    sh = [rnd(64, seed=20 + i) * 0.5 for i in range(2)]
    zb = torch.relu(yb * sc[0].view(1, 64, 1, 1) + sh[0].view(1, 64, 1, 1)).requires_grad_(True)
    zt = torch.relu(yt * sc[1].view(1, 64, 1, 1) + sh[1].view(1, 64, 1, 1)).requires_grad_(True)
    wb = rnd(64, 1, 2, 2, seed=3, scale=0.2).requires_grad_(True)
    wt = rnd(64, 1, 2, 2, seed=4, scale=0.2).requires_grad_(True)
    bb, bt = torch.tensor([0.1], requires_grad=True), torch.tensor([-0.2], requires_grad=True)
    P = torch.sigmoid(F.conv_transpose2d(zb, wb, bb, 2))
    T = torch.sigmoid(F.conv_transpose2d(zt, wt, bt, 2))
    ref = torch.cat([P, T, torch.reciprocal(1 + torch.exp(-50 * (P - T)))], 1)
    dpred = rnd(*ref.shape, seed=5)
    grads = torch.autograd.grad(ref, (zb, zt, wb, bb, wt, bt), dpred)
    d = lambda t: t.detach().contiguous().to(DEV)
    ybs, yts = nhwc(yb), nhwc(yt)
    wbd, wtd, bbd, btd = d(wb), d(wt), d(bb), d(bt)
    bn = [d(sc[0]), d(sh[0]), d(sc[1]), d(sh[1])]
    out = torch.full(ref.shape, float('nan'), device=DEV)
    _lib.check(L().dbn_head_tail_fwd(ybs.data_ptr(), yts.data_ptr(), wbd.data_ptr(), wtd.data_ptr(), bbd.data_ptr(), btd.data_ptr(),
                                     *[t.data_ptr() for t in bn], out.data_ptr(), N, Hq, Wq, 3, 50.0, stream()), 'head fwd')
    report('head tail fwd (fused BN+ReLU)', out.cpu(), ref, 2e-6, 1e-5)
    dxb, dxt = torch.empty_like(ybs), torch.empty_like(yts)
    dwb, dwt = torch.empty(256, device=DEV), torch.empty(256, device=DEV)
    dbb, dbt = torch.empty(1, device=DEV), torch.empty(1, device=DEV)
    ws = torch.empty(L().dbn_head_tail_bwd_ws_floats(), device=DEV)
    mean, rstd = [rnd(64, seed=30 + i) * 0.2 for i in range(2)], [rnd(64, seed=40 + i).abs() + 0.5 for i in range(2)]
    stats = [d(mean[0]), d(rstd[0]), d(mean[1]), d(rstd[1])]
    sums = torch.full((4, 64), float('nan'), device=DEV)
    _lib.check(L().dbn_head_tail_bwd(ybs.data_ptr(), yts.data_ptr(), wbd.data_ptr(), wtd.data_ptr(), out.data_ptr(), d(dpred).data_ptr(),
                                     *[t.data_ptr() for t in bn], *[t.data_ptr() for t in stats], sums.data_ptr(), dxb.data_ptr(),
                                     dxt.data_ptr(), dwb.data_ptr(), dbb.data_ptr(), dwt.data_ptr(), dbt.data_ptr(), N, Hq, Wq, 3, 50.0, 1.0,
                                     ws.data_ptr(), stream()), 'head bwd')
    # the fused per-channel reductions of the following BatchNorm backward: masked gradient and masked gradient * xhat
    for i, (yy, zz, g) in enumerate(((yb, zb, grads[0]), (yt, zt, grads[1]))):
        gm = (g * (zz > 0)).double()
        xhat = ((yy - mean[i].view(1, 64, 1, 1)) * rstd[i].view(1, 64, 1, 1)).double()
        report('fused BN sum g (branch %d)' % i, sums[2 * i].cpu(), gm.sum((0, 2, 3)).float(), 1e-4, 1e-4)
        report('fused BN sum g*xhat (branch %d)' % i, sums[2 * i + 1].cpu(), (gm * xhat).sum((0, 2, 3)).float(), 1e-4, 1e-4)
    report('fused head bwd dz_b', nchw(dxb), grads[0], 1e-4, 1e-3)
    report('fused head bwd dz_t', nchw(dxt), grads[1], 1e-4, 1e-3)
    report('fused head bwd dwb', dwb.cpu().view(64, 1, 2, 2), grads[2], 1e-3, 1e-3)
    report('fused head bwd dwt', dwt.cpu().view(64, 1, 2, 2), grads[4], 1e-3, 1e-3)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', [(2, 9, 7, True), (1, 16, 24, False), (3, 5, 13, True)])
def test_db_head_tail_of_both_branches_in_one_launch_on_16bit_storage(case, dtype):
    """dbn_head16_tail_eval_t (round 5, inference): per branch ConvTranspose2d(64, 64, 2, 2) -> eval-mode BatchNorm -> ReLU ->
    ConvTranspose2d(64, 1, 2, 2) -> Sigmoid (segmentation_head.py:27-29,35-45,74-79), both branches, one launch, the two 64-channel
    half-resolution tensors never written.  Reference: fp64 on the operands as stored (input, first weight and second weight rounded to the
    storage type; the kernel also rounds the activations between the two ConvTs: that is the tolerance); and the three-kernel path
    (dbn_convt16_bn_t twice + dbn_head_tail_fwd_t).  Ragged last 32-pixel block."""
    N, Hq, Wq, with_bias = case
    at = AT_OF[dtype]
    eps = 2.0**-8 if dtype == torch.bfloat16 else 2.0**-10
    rq = lambda t: t.to(dtype).double()
    xs, w1, b1, sc, sh, w2, b2 = [], [], [], [], [], [], []
    refs, logits = [], []
    for i in range(2):
        x = rnd(N, 64, Hq, Wq, seed=1 + i).abs()  # (post-ReLU activations)
        xs.append(x)
        w1.append(rnd(64, 64, 2, 2, seed=3 + i, scale=0.15))
        b1.append(rnd(64, seed=5 + i) * 0.1)
        sc.append(rnd(64, seed=7 + i) * 0.3 + 1)
        sh.append(rnd(64, seed=9 + i) * 0.5)
        w2.append(rnd(64, 1, 2, 2, seed=11 + i, scale=0.2))
        b2.append(torch.tensor([0.1 - 0.3 * i]))
        y = F.conv_transpose2d(rq(x), rq(w1[i]), b1[i].double() if with_bias else None, 2)
        z = torch.relu(y * sc[i].double().view(1, 64, 1, 1) + sh[i].double().view(1, 64, 1, 1))
        lg = F.conv_transpose2d(z, rq(w2[i]), b2[i].double(), 2)
        logits.append(lg)
        refs.append(torch.sigmoid(lg))
    ref = torch.cat(refs, 1)
    d = lambda t: t.contiguous().to(DEV)
    xd = [nhwc(x).to(dtype) for x in xs]
    panels = []
    for i in range(2):
        pnl = torch.empty(L().dbn_convt16_panel_bytes(), device=DEV, dtype=torch.uint8)
        w1d = d(w1[i])
        _lib.check(L().dbn_convt16_pack(at, w1d.data_ptr(), pnl.data_ptr(), stream()), 'convt16_pack')
        panels.append(pnl)
    b1d, scd, shd, w2d, b2d = [d(t) for t in b1], [d(t) for t in sc], [d(t) for t in sh], [d(t) for t in w2], [d(t) for t in b2]
    assert L().dbn_head16_eligible(at, N, Hq, Wq) == 1
    out = torch.full((N, 2, 4 * Hq, 4 * Wq), float('nan'), device=DEV)
    bp = lambda i: b1d[i].data_ptr() if with_bias else None
    _lib.check(L().dbn_head16_tail_eval_t(at, xd[0].data_ptr(), xd[1].data_ptr(), panels[0].data_ptr(), panels[1].data_ptr(), bp(0), bp(1),
                                          scd[0].data_ptr(), shd[0].data_ptr(), scd[1].data_ptr(), shd[1].data_ptr(), w2d[0].data_ptr(),
                                          w2d[1].data_ptr(), b2d[0].data_ptr(), b2d[1].data_ptr(), out.data_ptr(), N, Hq, Wq, stream()), 'head16')
    lscale = max(float(l.abs().max()) for l in logits)
    report('head tail in one launch %s %s' % (case, dtype), out.cpu(), ref, 0.25 * 3 * eps * (lscale + 1.0), 0.0)
    # the three-kernel path on the same operands (rounds the first ConvT's output instead of the activation, and keeps the second weight in fp32)
    ys = []
    for i in range(2):
        y1 = torch.empty((N, 2 * Hq, 2 * Wq, 64), device=DEV, dtype=dtype)
        _lib.check(L().dbn_convt16_bn_t(at, xd[i].data_ptr(), panels[i].data_ptr(), bp(i), y1.data_ptr(), N, Hq, Wq, None, None, 0.0, 0.0, None,
                                        None, None, None, None, None, None, stream()), 'convt16')
        ys.append(y1)
    out3 = torch.full_like(out, float('nan'))
    _lib.check(L().dbn_head_tail_fwd_t(at, ys[0].data_ptr(), ys[1].data_ptr(), w2d[0].data_ptr(), w2d[1].data_ptr(), b2d[0].data_ptr(),
                                       b2d[1].data_ptr(), scd[0].data_ptr(), shd[0].data_ptr(), scd[1].data_ptr(), shd[1].data_ptr(),
                                       out3.data_ptr(), N, 2 * Hq, 2 * Wq, 2, 50.0, stream()), 'head_tail_fwd')
    report('one launch vs three', out.cpu(), out3.cpu(), 0.25 * 5 * eps * (lscale + 1.0), 0.0)
    out2 = torch.full_like(out, float('nan'))
    _lib.check(L().dbn_head16_tail_eval_t(at, xd[0].data_ptr(), xd[1].data_ptr(), panels[0].data_ptr(), panels[1].data_ptr(), bp(0), bp(1),
                                          scd[0].data_ptr(), shd[0].data_ptr(), scd[1].data_ptr(), shd[1].data_ptr(), w2d[0].data_ptr(),
                                          w2d[1].data_ptr(), b2d[0].data_ptr(), b2d[1].data_ptr(), out2.data_ptr(), N, Hq, Wq, stream()), 'head16')
    assert torch.equal(out, out2)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('ch', [2, 3])
def test_head_tail_forward_lane_layouts_on_16bit_storage(dtype, ch):
    """dbn_head_tail_fwd_t on bf16 / fp16 inputs (segmentation_head.py:27-29,35-45,74-79 with the BatchNorm + ReLU of the ConvT outputs applied
    on load): the eight-lanes-per-pixel form that large maps take (round 5, 16-byte loads) against the sixteen-lane form and against fp64 on
    the inputs as stored."""
    N, Hq, Wq = 2, 11, 13
    at = AT_OF[dtype]
    rq = lambda t: t.to(dtype).double()
    yb, yt = rnd(N, 64, Hq, Wq, seed=1), rnd(N, 64, Hq, Wq, seed=2)
    sc = [rnd(64, seed=10 + i) * 0.3 + 1 for i in range(2)]
    sh = [rnd(64, seed=20 + i) * 0.5 for i in range(2)]
    zb = torch.relu(rq(yb) * sc[0].double().view(1, 64, 1, 1) + sh[0].double().view(1, 64, 1, 1))
    zt = torch.relu(rq(yt) * sc[1].double().view(1, 64, 1, 1) + sh[1].double().view(1, 64, 1, 1))
    wb, wt = rnd(64, 1, 2, 2, seed=3, scale=0.2), rnd(64, 1, 2, 2, seed=4, scale=0.2)
    bb, bt = torch.tensor([0.1]), torch.tensor([-0.2])
    P = torch.sigmoid(F.conv_transpose2d(zb, wb.double(), bb.double(), 2))
    T = torch.sigmoid(F.conv_transpose2d(zt, wt.double(), bt.double(), 2))
    ref = torch.cat([P, T] + ([torch.reciprocal(1 + torch.exp(-50 * (P - T)))] if ch == 3 else []), 1)
    d = lambda t: t.contiguous().to(DEV)
    ybs, yts = nhwc(yb).to(dtype), nhwc(yt).to(dtype)
    args = [d(wb), d(wt), d(bb), d(bt), d(sc[0]), d(sh[0]), d(sc[1]), d(sh[1])]
    outs = []
    old = L().dbn_set_head_tail_wide(0)
    try:
        for mode in (-1, 1):
            L().dbn_set_head_tail_wide(mode)
            out = torch.full(ref.shape, float('nan'), device=DEV)
            _lib.check(L().dbn_head_tail_fwd_t(at, ybs.data_ptr(), yts.data_ptr(), *[t.data_ptr() for t in args], out.data_ptr(), N, Hq, Wq, ch,
                                               50.0, stream()), 'head fwd')
            outs.append(out.cpu())
    finally:
        L().dbn_set_head_tail_wide(old)
    # (the step function's slope of 50 amplifies the fp32 rounding of the two logits: the binary map gets the wider bound)
    for i, name in enumerate(('sixteen lanes', 'eight lanes')):
        report('head tail fwd %s %s P, T' % (name, dtype), outs[i][:, :2], ref[:, :2], 2e-6, 1e-5)
        if ch == 3:
            report('head tail fwd %s %s B' % (name, dtype), outs[i][:, 2:], ref[:, 2:], 1e-4, 1e-4)
    report('eight lanes vs sixteen', outs[1][:, :2], outs[0][:, :2], 1e-6, 1e-6)


@pytest.mark.parametrize('ns', [0, 3])
def test_batched_weight_pack_equals_single_packs(ns):
    """dbn_pack_weights_batched (one launch, device job table) produces exactly the panels of the per-weight
    dbn_pack_weights / dbn_pack_weights_bf16s calls, including the f*f parity-class panels of strided transposed convs."""
    import ctypes
    jobs_spec = [((64, 3, 7, 7), 0, 2), ((128, 64, 3, 3), 0, 1), ((128, 64, 3, 3), 1, 2), ((64, 256, 10, 10), 1, 8), ((64, 64, 2, 2), 1, 2),
                 ((256, 128, 1, 1), 1, 1), ((64, 256, 6, 6), 1, 4)]

    class Job(ctypes.Structure):
        _fields_ = [('w', ctypes.c_void_p), ('out', ctypes.c_void_p)] + [(f, ctypes.c_int) for f in ('O', 'I', 'R', 'S', 'mode', 'Cs', 'Cd', 'f')]

    ws, singles, outs = [], [], []
    arr = (Job * len(jobs_spec))()
    for i, (shape, mode, stride) in enumerate(jobs_spec):
        w = rnd(*shape, seed=50 + i)
        O, I, R, S = shape
        if (mode == 0 and O % 64) or (mode == 1 and I % 64):
            continue
        wd = w.to(DEV)
        singles.append(pack(w, mode, stride, ns))
        out = torch.full_like(singles[-1], float('nan'))
        ws.append(wd)
        outs.append(out)
        arr[len(outs) - 1] = Job(wd.data_ptr(), out.data_ptr(), O, I, R, S, mode, (I + 3) // 4 * 4 if mode == 0 else O, O if mode == 0 else I,
                                 stride if (mode == 1 and stride > 1) else 1)
    n = len(outs)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)[:n * ctypes.sizeof(Job)].clone().to(DEV)
    _lib.check(L().dbn_pack_weights_batched(table.data_ptr(), n, ns, stream()), 'pack_batched')
    torch.cuda.synchronize()
    for a, b in zip(outs, singles):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_image_chunking_beyond_the_index_ranges():
    """The conv / weight-gradient kernels index pixels in 24 bits and address tensors with 32-bit offsets; a call beyond
    those ranges (e.g. BASELINE configs[4] at batch sizes above 40: 48 x 1280^2 / 4 px = 19.7 M rows) runs as several
    launches over image ranges.  With the limits lowered (dbn_set_index_limits) the chunked path must reproduce the
    single-launch results BIT FOR BIT: conv + fused BN statistics, strided transposed conv, pyramid conv, weight gradient."""
    N, Ci, Co, H, W = 5, 64, 128, 24, 16
    x = nhwc(rnd(N, Ci, H, W, seed=1))
    w = rnd(Co, Ci, 3, 3, seed=2, scale=0.05)
    wT = rnd(Ci, Co, 4, 4, seed=3, scale=0.05)
    bias = rnd(Co, seed=4).to(DEV)
    dy = nhwc(rnd(N, Co, H, W, seed=5))

    def run_all(convT_limit=0):
        out = {}
        y = torch.full((N, H, W, Co), float('nan'), device=DEV)
        g_, b_ = torch.ones(Co, device=DEV), torch.zeros(Co, device=DEV)
        rm_, rv_ = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
        sc, sh, mu, rs = (torch.empty(Co, device=DEV) for _ in range(4))
        ws = torch.empty(L().dbn_conv_bn_ws_floats(N, H, W, Co, 0, 1), device=DEV)
        _lib.check(L().dbn_conv_bn_f32(x.data_ptr(), pack(w, 0).data_ptr(), bias.data_ptr(), y.data_ptr(), N, H, W, Ci, H, W, Co, 3, 3, 1, 1,
                                       0, 0, 0, 0, g_.data_ptr(), b_.data_ptr(), 1e-5, 0.1, rm_.data_ptr(), rv_.data_ptr(), sc.data_ptr(),
                                       sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), ws.data_ptr(), stream()), 'conv_bn')
        out['conv'], out['scale'], out['shift'], out['run_var'] = y, sc, sh, rv_
        yT = torch.full((N, 2 * H, 2 * W, Co), float('nan'), device=DEV)
        if convT_limit:  # its output has 4x the pixels: one image must still fit a launch
            L().dbn_set_index_limits(convT_limit, 0, 0)
        igemm(x, pack(wT, 1, 2), bias, yT, 4, 2, 1, 1)
        if convT_limit:
            L().dbn_set_index_limits(2 * H * W + 7, 0, 0)
        out['convT'] = yT
        d = torch.full((N, H, W, Ci), float('nan'), device=DEV)
        igemm(dy, pack(w, 1, 1), None, d, 3, 1, 1, 1)
        out['dgrad'] = d
        out['wgrad'] = wgrad(dy, x, Co, Ci, 3, 1, 1)
        # pyramid conv over four levels (H, W multiples of 8)
        Cg = 64
        zs = [nhwc(rnd(N, Cg, H >> g, W >> g, seed=20 + g)) for g in range(4)]
        wf = rnd(256, 4 * Cg, 3, 3, seed=6, scale=0.03).to(DEV)
        wpk = []
        for g in range(4):
            k = (1 << g) + 2
            wdg = torch.empty(Cg, 256, k, k, device=DEV)
            _lib.check(L().dbn_fpn_combine_weights(wf.data_ptr(), 256, 4 * Cg, g, Cg, wdg.data_ptr(), stream()), 'combine')
            wpk.append(pack(wdg.cpu(), 1, 1 << g))
        yp = torch.full((N, H, W, 256), float('nan'), device=DEV)
        sc2, sh2, mu2, rs2 = (torch.empty(256, device=DEV) for _ in range(4))
        g2, b2 = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
        ws2 = torch.empty(L().dbn_pyramid_conv_ws_floats(N, H, W, 256), device=DEV)
        _lib.check(L().dbn_pyramid_conv_f32(*[z.data_ptr() for z in zs], *[p.data_ptr() for p in wpk], None, yp.data_ptr(), N, H, W, Cg, 256,
                                            0, 0, g2.data_ptr(), b2.data_ptr(), 1e-5, 0.1, None, None, sc2.data_ptr(), sh2.data_ptr(),
                                            mu2.data_ptr(), rs2.data_ptr(), ws2.data_ptr(), stream()), 'pyramid')
        out['pyramid'], out['pyramid_scale'] = yp, sc2
        torch.cuda.synchronize()
        return out

    ref = run_all()
    assert L().dbn_wgrad_splitk_hw(N, H, W, Co, H, W, Ci, 3, 3) == L().dbn_wgrad_splitk(N, H, W, Co, Ci, 3, 3)
    try:
        L().dbn_set_index_limits(2 * H * W + 7, 0, 0)  # at most two images per launch
        assert L().dbn_wgrad_splitk_hw(N, H, W, Co, H, W, Ci, 3, 3) >= 3
        got = run_all(convT_limit=8 * H * W + 7)
        L().dbn_set_index_limits(0, H * W * Co * 4 + 64, 0)  # one image of the widest tensor per launch, through the byte range
        got1 = run_all()
    finally:
        L().dbn_set_index_limits(0, 0, 0)
    for k, v in ref.items():
        if k == 'wgrad':  # the pixel splits (and their fp64 fold order) differ with the chunking: same sum, other rounding
            report('chunked wgrad', got[k].cpu(), v.cpu(), 1e-5 * float(v.abs().max()), 1e-5)
            report('chunked wgrad (1/launch)', got1[k].cpu(), v.cpu(), 1e-5 * float(v.abs().max()), 1e-5)
        elif k in ('scale', 'shift', 'run_var', 'pyramid_scale'):  # tile partials are merged in fp64: identical up to the last bit
            report('chunked ' + k, got[k].cpu(), v.cpu(), 1e-7, 1e-6)
            report('chunked ' + k + ' (1/launch)', got1[k].cpu(), v.cpu(), 1e-7, 1e-6)
        else:
            assert torch.equal(got[k], v), k
            assert torch.equal(got1[k], v), k


AT_OF = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def igemm_t(src, wpk, bias, dst, R, stride, pad, mode, accumulate=0, tile=0, ns=1, ksplit=1, slab=None):
    N, Hs, Ws, Cs = src.shape
    _, Hd, Wd, Cd = dst.shape
    _lib.check(L().dbn_igemm_t(AT_OF[src.dtype], ns, src.data_ptr(), wpk.data_ptr(), None if bias is None else bias.data_ptr(), dst.data_ptr(),
                               N, Hs, Ws, Cs, Hd, Wd, Cd, R, R, stride, pad, mode, accumulate, tile, ksplit,
                               None if slab is None else slab.data_ptr(), stream()), 'igemm_t')


def pack_t(w, mode, stride, kind, cs=0):
    O, I, R, S = w.shape
    wd = w.contiguous().to(DEV)
    out = torch.empty(L().dbn_igemm_panel_floats_t(kind, O, I, R, S, mode, stride, cs), device=DEV)
    _lib.check(L().dbn_pack_weights_t(kind, wd.data_ptr(), O, I, R, S, mode, stride, cs, out.data_ptr(), stream()), 'pack_t')
    return out


@pytest.mark.parametrize('case', [(8, 200, 64, 256, 1, 1, 0, 0), (8, 200, 64, 256, 1, 1, 0, 1), (8, 200, 128, 128, 3, 2, 1, 1), (8, 50, 2304, 256, 1, 1, 0, 1),
                                  (8, 100, 128, 64, 3, 1, 1, 1), (2, 24, 2048, 512, 1, 1, 0, 0), (8, 25, 4608, 512, 1, 1, 0, 0)])
def test_16bit_generic_loop_on_long_launches(case):
    """The 16-bit generic loop keeps several A stages (LDS-DMA) and a weight-fragment set (register loads) in flight behind COUNTED
    vmcnt waits; a count that does not match what is really outstanding reads a stage before it has landed — a race that only long
    launches lose (configs[3] shapes: resnet50 / deformable at 800 x 800; the first version of the register-fragment form passed
    every small-shape test and failed here).  bf16 storage, against F.conv2d / conv_transpose2d on the same rounded operands; three
    runs must agree bit for bit."""
    N, H, Cs, Cd, k, s, p, mode = case
    x = (rnd(N, Cs, H, H, seed=1)).to(torch.bfloat16)
    if mode == 0:
        w = (rnd(Cd, Cs, k, k, seed=2, scale=(1.0 / (Cs * k * k))**0.5)).to(torch.bfloat16)
        ref = F.conv2d(x.float(), w.float(), None, s, p)
    else:
        w = (rnd(Cs, Cd, k, k, seed=2, scale=(1.0 / (Cs * k * k))**0.5)).to(torch.bfloat16)
        ref = F.conv_transpose2d(x.float(), w.float(), None, s, p, output_padding=s - 1 if s > 1 else 0)
    Hd = ref.shape[2]
    xs = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp = pack_t(w.float(), mode, s, 1, Cs if mode == 0 else 0)
    outs = []
    for _ in range(3):
        y = torch.full((N, Hd, Hd, Cd), float('nan'), device=DEV, dtype=torch.bfloat16)
        igemm_t(xs, wp, None, y, k, s, p, mode)
        outs.append(y)
    got = outs[0].float().permute(0, 3, 1, 2).cpu()
    scale = float(ref.abs().max())
    report('16-bit generic loop %s' % (case, ), got, ref, 2.0**-7 * scale, 2.0**-7)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('tile', [0, 1, 2, 4])
def test_convolutions_on_16bit_storage(dtype, tile):
    """dbn_igemm_t with bf16 / fp16 activation storage (BASELINE configs[2]/[4]): the source is read as stored (8-channel
    pieces straight into the LDS image), the weights come from bf16 / fp16 panels, products accumulate in fp32 and the result is
    rounded once to the storage type.  Reference: F.conv2d / conv_transpose2d in fp64 on the SAME rounded operands; the only
    error left is the output rounding (2^-9 bf16, 2^-11 fp16, relative) plus fp32 accumulation noise."""
    kind = 1 if dtype == torch.bfloat16 else 2
    eps = 2.0**-8 if dtype == torch.bfloat16 else 2.0**-10
    rq = lambda t: t.to(dtype).double()  # rounded to storage, as fp64
    N, Ci, Co, H, W = 3, 64, 128, 14, 18
    x = rnd(N, Ci, H, W, seed=1)
    xs = nhwc(x).to(dtype)
    b = rnd(Co, seed=3)
    for (k, s_, p_) in ((3, 1, 1), (1, 1, 0), (3, 2, 1)):
        w = rnd(Co, Ci, k, k, seed=2, scale=(2.0 / (Ci * k * k))**0.5)
        ref = F.conv2d(rq(x), rq(w), b.double(), s_, p_)
        y = torch.full((N, ref.shape[2], ref.shape[3], Co), float('nan'), device=DEV, dtype=dtype)
        igemm_t(xs, pack_t(w, 0, s_, kind, Ci), b.to(DEV), y, k, s_, p_, 0, tile=tile)
        report('conv k%d s%d %s' % (k, s_, dtype), nchw(y.float()), ref, eps * float(ref.abs().max()) * 0.5, eps)
    # data gradient (mode 1, stride 1, accumulate) and ConvTranspose2d (mode 1, stride 2: parity classes)
    w = rnd(Co, Ci, 3, 3, seed=4, scale=0.05)
    dy = rnd(N, Co, H, W, seed=5)
    base = rnd(N, Ci, H, W, seed=6)
    ref = rq(base) + F.conv_transpose2d(rq(dy), rq(w), None, 1, 1)
    d = nhwc(base).to(dtype)
    igemm_t(nhwc(dy).to(dtype), pack_t(w, 1, 1, kind), None, d, 3, 1, 1, 1, accumulate=1, tile=tile)
    report('dgrad+acc %s' % dtype, nchw(d.float()), ref, eps * float(ref.abs().max()) * 0.5, eps)
    wT = rnd(Ci, Co, 2, 2, seed=7, scale=0.1)
    ref = F.conv_transpose2d(rq(x), rq(wT), b.double(), 2, 0)
    yT = torch.full((N, 2 * H, 2 * W, Co), float('nan'), device=DEV, dtype=dtype)
    igemm_t(xs, pack_t(wT, 1, 2, kind), b.to(DEV), yT, 2, 2, 0, 1, tile=tile)
    report('convT k2 s2 %s' % dtype, nchw(yT.float()), ref, eps * float(ref.abs().max()) * 0.5, eps)
    # split-K (fp32 slabs, one rounding in the slab sum)
    w = rnd(Co, Ci, 3, 3, seed=8, scale=0.05)
    ref = F.conv2d(rq(x), rq(w), b.double(), 1, 1)
    y = torch.full((N, H, W, Co), float('nan'), device=DEV, dtype=dtype)
    slab = torch.empty(4 * (y.numel() + 1088), device=DEV)
    igemm_t(xs, pack_t(w, 0, 1, kind, Ci), b.to(DEV), y, 3, 1, 1, 0, tile=tile, ksplit=4, slab=slab)
    report('split-K conv %s' % dtype, nchw(y.float()), ref, eps * float(ref.abs().max()) * 0.5, eps)


@pytest.fixture(params=[0, 3], ids=['tile128x128', 'tile128x256'])
def pyramid_wide(request):
    """Both tiles of the 16-bit pyramid conv (round 6: 128 x 256 where Cd % 256 == 0 — dbn_set_pyramid_wide, 3 = whatever the launch's size; other
    Cd: the 128 x 128 tile either way)."""
    old = L().dbn_set_pyramid_wide(request.param)
    yield request.param
    L().dbn_set_pyramid_wide(old)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(2, 16, 32, 64, 256), (3, 40, 16, 32, 128)])
@pytest.mark.parametrize('first', [0, 1])
def test_pyramid_conv_on_16bit_storage_from_level_one(shape, dtype, first, pyramid_wide):
    """The pyramid conv on bf16 / fp16 storage, whole (first_level = 0) and as levels 1-3 accumulated onto level 0's part, which a mode-1
    3x3 launch on the level-0 panel wrote first (round 5: that launch takes the pixel-patch kernel).  Reference: the concat conv in fp64 on
    the rounded operands (segmentation_body.py:75-76,82-87); first_level = 1 pays one more rounding of the level-0 part to the storage type.
    Also the inference epilogue (ReLU) and the train-mode statistics of the accumulated form."""
    N, H, W, Cg, Co = shape
    kind = AT_OF[dtype]
    eps = 2.0**-8 if dtype == torch.bfloat16 else 2.0**-10
    rq = lambda t: t.to(dtype).double()
    ps = [rnd(N, Cg, H >> g, W >> g, seed=20 + g) for g in range(4)]
    w = rnd(Co, 4 * Cg, 3, 3, seed=3, scale=0.03)
    bias = rnd(Co, seed=5)
    wd_ = w.to(DEV)
    wpk = []
    for g in range(4):
        k = (1 << g) + 2
        wdg = torch.empty(Cg, Co, k, k, device=DEV)
        _lib.check(L().dbn_fpn_combine_weights(wd_.data_ptr(), Co, 4 * Cg, g, Cg, wdg.data_ptr(), stream()), 'combine')
        wpk.append(pack_t(wdg.cpu(), 1, 1 << g, kind))
    cat = torch.cat([rq(ps[0])] + [F.interpolate(rq(ps[g]), size=(H, W)) for g in range(1, 4)], 1)
    y_ref = F.conv2d(cat, w.double(), bias.double(), 1, 1)
    xs = [nhwc(t).to(dtype) for t in ps]
    bias_ = bias.to(DEV)
    scale = float(y_ref.abs().max())
    # (the combined filters are sums of up to 9 taps rounded ONCE to the storage type: allow that on top of the output rounding)
    tol_a, tol_r = eps * scale * (1.5 if first == 0 else 2.5), 2 * eps
    for relu in (0, 1):
        y = torch.full((N, H, W, Co), float('nan'), device=DEV, dtype=dtype)
        if first:
            igemm_t(xs[0], wpk[0], bias_, y, 3, 1, 1, 1)
        _lib.check(L().dbn_pyramid_conv_act_t(first, kind, *[t.data_ptr() for t in xs], *[t.data_ptr() for t in wpk], bias_.data_ptr(), relu,
                                              y.data_ptr(), N, H, W, Cg, Co, 1, stream()), 'pyramid_act')
        report('pyramid16 first=%d relu=%d %s' % (first, relu, dtype), nchw(y.float()), y_ref.clamp_min(0) if relu else y_ref, tol_a, tol_r)
    # train-mode statistics of the same sum
    gamma, beta = rnd(Co, seed=6) * 0.3 + 1, rnd(Co, seed=7)
    g_, b_ = gamma.to(DEV), beta.to(DEV)
    rm_, rv_ = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
    sc, sh, mu, rs = (torch.empty(Co, device=DEV) for _ in range(4))
    ws = torch.empty(L().dbn_pyramid_conv_ws_floats(N, H, W, Co), device=DEV)
    y = torch.full((N, H, W, Co), float('nan'), device=DEV, dtype=dtype)
    if first:
        igemm_t(xs[0], wpk[0], bias_, y, 3, 1, 1, 1)
    _lib.check(L().dbn_pyramid_conv_from_t(first, kind, *[t.data_ptr() for t in xs], *[t.data_ptr() for t in wpk], bias_.data_ptr(), y.data_ptr(),
                                           N, H, W, Cg, Co, 0, 1, g_.data_ptr(), b_.data_ptr(), 1e-5, 0.1, rm_.data_ptr(), rv_.data_ptr(),
                                           sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), ws.data_ptr(), stream()), 'pyramid_from')
    report('pyramid16 train y first=%d' % first, nchw(y.float()), y_ref, tol_a, tol_r)
    mean_ref = y_ref.mean((0, 2, 3))
    var_ref = y_ref.var((0, 2, 3), unbiased=False)
    report('pyramid16 batch mean', mu.cpu(), mean_ref, tol_a, tol_r)
    report('pyramid16 batch rstd', rs.cpu(), (var_ref + 1e-5).rsqrt(), 0.0, 4 * eps)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', [(3, 12, 20, 64, 256, 3, 1, 1, 0), (3, 12, 20, 256, 256, 3, 1, 1, 1), (2, 16, 16, 128, 512, 1, 1, 0, 0), (2, 18, 22, 64, 256, 3, 2, 1, 0),
                                  (8, 50, 50, 512, 512, 3, 1, 1, 0)])
def test_generic_16bit_convs_on_the_wide_tile_agree_bit_for_bit(case, dtype):
    """dbn_set_pyramid_wide(3): plain forward / stride-1 data-gradient launches of the generic 16-bit loop on the 128 x 256 tile against the
    128 x 128 tile (0): equal bits (same products, same order), bias / accumulate included; the long 512 -> 512 case is the counted-wait race
    screen of test_16bit_generic_loop_on_long_launches at this tile."""
    N, H, W, Ci, Co, k, st_, pad, mode = case
    kind = AT_OF[dtype]
    x = nhwc(rnd(N, Ci, H, W, seed=31)).to(dtype)
    Hd, Wd = ((H + 2 * pad - k) // st_ + 1, (W + 2 * pad - k) // st_ + 1) if mode == 0 else (H, W)
    w = rnd(Co, Ci, k, k, seed=32, scale=(1.0 / (Ci * k * k))**0.5) if mode == 0 else rnd(Ci, Co, k, k, seed=32, scale=(1.0 / (Ci * k * k))**0.5)
    wp = pack_t(w, mode, 1, kind, Ci if mode == 0 else 0)
    b = rnd(Co, seed=33).to(DEV)
    base = nhwc(rnd(N, Co, Hd, Wd, seed=34)).to(dtype)

    def run(wide, accumulate):
        old = L().dbn_set_pyramid_wide(wide)
        try:
            y = base.clone() if accumulate else torch.full((N, Hd, Wd, Co), float('nan'), device=DEV, dtype=dtype)
            igemm_t(x, wp, None if accumulate else b, y, k, st_, pad, mode, accumulate=accumulate)
            torch.cuda.synchronize()
            return y
        finally:
            L().dbn_set_pyramid_wide(old)

    for accumulate in (0, 1):
        a, c, c2 = run(0, accumulate), run(3, accumulate), run(3, accumulate)
        assert torch.isfinite(a.float()).all()
        assert torch.equal(c, c2), 'wide tile: runs differ'
        assert torch.equal(a, c), 'wide tile differs from the 128 x 128 tile (accumulate=%d)' % accumulate
    if mode == 0:
        ref = F.conv2d(nchw(x.double()), w.to(dtype).double(), b.double().cpu(), st_, pad)
        eps = 2.0**-8 if dtype == torch.bfloat16 else 2.0**-10
        report('wide tile vs fp64', nchw(run(3, 0).float()), ref, eps * float(ref.abs().max()) * 0.5, eps)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(2, 16, 32, 64, 256), (5, 24, 40, 32, 512)])
def test_pyramid_conv_tiles_agree_bit_for_bit(shape, dtype):
    """The 128 x 256 tile of the 16-bit pyramid conv sums the same products in the same order as the 128 x 128 tile: equal outputs, train-mode
    BatchNorm statistics equal to fp32 summation order (ragged last row tile: 5 x 3 x 5 = 75 blocks of 8 x 8; two column tiles at Co = 512)."""
    N, H, W, Cg, Co = shape
    kind = AT_OF[dtype]
    xs = [nhwc(rnd(N, Cg, H >> g, W >> g, seed=40 + g)).to(dtype) for g in range(4)]
    wd_ = rnd(Co, 4 * Cg, 3, 3, seed=9, scale=0.03).to(DEV)
    wpk = []
    for g in range(4):
        k = (1 << g) + 2
        wdg = torch.empty(Cg, Co, k, k, device=DEV)
        _lib.check(L().dbn_fpn_combine_weights(wd_.data_ptr(), Co, 4 * Cg, g, Cg, wdg.data_ptr(), stream()), 'combine')
        wpk.append(pack_t(wdg.cpu(), 1, 1 << g, kind))
    bias_ = rnd(Co, seed=5).to(DEV)
    g_, b_ = (rnd(Co, seed=6) * 0.3 + 1).to(DEV), rnd(Co, seed=7).to(DEV)

    def run(wide):
        old = L().dbn_set_pyramid_wide(wide)
        try:
            ya = torch.full((N, H, W, Co), float('nan'), device=DEV, dtype=dtype)
            _lib.check(L().dbn_pyramid_conv_act_t(0, kind, *[t.data_ptr() for t in xs], *[t.data_ptr() for t in wpk], bias_.data_ptr(), 1,
                                                  ya.data_ptr(), N, H, W, Cg, Co, 1, stream()), 'pyramid_act')
            rm_, rv_ = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
            sc, sh, mu, rs = (torch.empty(Co, device=DEV) for _ in range(4))
            ws = torch.empty(L().dbn_pyramid_conv_ws_floats(N, H, W, Co), device=DEV)
            yt = torch.full((N, H, W, Co), float('nan'), device=DEV, dtype=dtype)
            _lib.check(L().dbn_pyramid_conv_from_t(0, kind, *[t.data_ptr() for t in xs], *[t.data_ptr() for t in wpk], bias_.data_ptr(), yt.data_ptr(),
                                                   N, H, W, Cg, Co, 0, 1, g_.data_ptr(), b_.data_ptr(), 1e-5, 0.1, rm_.data_ptr(), rv_.data_ptr(),
                                                   sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), ws.data_ptr(), stream()), 'pyramid_from')
            torch.cuda.synchronize()
            return dict(act=ya, train=yt, scale=sc, shift=sh, mean=mu, rstd=rs, run_mean=rm_, run_var=rv_)
        finally:
            L().dbn_set_pyramid_wide(old)

    a, b = run(0), run(3)
    assert torch.isfinite(a['act'].float()).all() and torch.isfinite(a['train'].float()).all()
    for k in a:
        if k in ('act', 'train'):
            assert torch.equal(a[k], b[k]), k
        else:  # the statistics: a wave of the wide tile sums its 128 rows in another order (1 x 4 waves of 128 x 64 against 2 x 2 of 64 x 64)
            report('%s, wide vs 128 x 128 tile' % k, b[k].cpu().double(), a[k].cpu().double(), 2e-6 * float(a[k].abs().max()), 2e-6)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', [(2, 16, 24, True, True), (1, 7, 9, False, True), (3, 5, 13, True, False)])
@pytest.mark.parametrize('Co', [64, 256])
def test_pointwise_conv_from_64_channels_on_16bit_storage(case, dtype, Co):
    """dbn_pw16_act_t (round 5): nn.Conv2d(64, 64 | 256, 1) + bias + ReLU on bf16 / fp16 storage — the FPN lateral on c2 in inference
    (segmentation_body.py:46,68 with the eval-mode BatchNorm folded, basic.py:32-36) on csrc/convt16.hip's kernel.  Reference: fp64 on the
    operands as stored; the only error left is the output rounding.  Ragged last block (N * H * W not a multiple of 32)."""
    N, H, W, with_bias, relu = case
    at = AT_OF[dtype]
    eps = 2.0**-8 if dtype == torch.bfloat16 else 2.0**-10
    rq = lambda t: t.to(dtype).double()
    assert L().dbn_pw16_eligible(at, N, H, W, 64, Co) == 1 and L().dbn_pw16_eligible(at, N, H, W, 128, Co) == 0
    assert L().dbn_pw16_eligible(at, N, H, W, 64, 128) == 0
    x = rnd(N, 64, H, W, seed=1)
    w = rnd(Co, 64, 1, 1, seed=2, scale=0.2)
    b = rnd(Co, seed=3)
    ref = F.conv2d(rq(x), rq(w), b.double() if with_bias else None)
    if relu:
        ref = ref.clamp_min(0)
    panel = torch.empty(L().dbn_pw16_panel_bytes(), device=DEV, dtype=torch.uint8)
    wd, bd = w.to(DEV), b.to(DEV)
    _lib.check(L().dbn_pw16_pack(at, wd.data_ptr(), Co, panel.data_ptr(), stream()), 'pw16_pack')
    xs = nhwc(x).to(dtype)
    y = torch.full((N, H, W, Co), float('nan'), device=DEV, dtype=dtype)
    _lib.check(L().dbn_pw16_act_t(at, xs.data_ptr(), panel.data_ptr(), bd.data_ptr() if with_bias else None, int(relu), y.data_ptr(), N, H, W,
                                  Co, stream()), 'pw16')
    report('pointwise 64->%d %s %s' % (Co, case, dtype), nchw(y.float()), ref, eps * float(ref.abs().max()) * 0.5, eps)
    # the generic launch on the same operands: identical products, fp32 accumulation in a different order
    y2 = torch.full((N, H, W, Co), float('nan'), device=DEV, dtype=dtype)
    _lib.check(L().dbn_igemm_act_t(at, 1, xs.data_ptr(), pack_t(w, 0, 1, at, 64).data_ptr(), bd.data_ptr() if with_bias else None, None, int(relu),
                                   y2.data_ptr(), N, H, W, 64, H, W, Co, 1, 1, 1, 0, 0, 0, stream()), 'igemm_act')
    report('pointwise vs generic launch', y.float().cpu(), y2.float().cpu(), eps * float(ref.abs().max()), eps)


def test_stem_conv_on_16_channel_bf16_input():
    """The 16-bit path stores the model input with 16 channels (3 real): nchw3_to_nhwc4_t + the 7x7 stride-2 stem conv."""
    N, H, W = 2, 40, 48
    x = rnd(N, 3, H, W, seed=1)
    w = rnd(64, 3, 7, 7, seed=2, scale=0.1)
    xd = x.to(DEV)
    x16 = torch.full((N, H, W, 16), float('nan'), device=DEV, dtype=torch.bfloat16)
    _lib.check(L().dbn_nchw3_to_nhwc4_t(1, xd.data_ptr(), x16.data_ptr(), N, H, W, stream()), 'nchw3_to_nhwc16')
    assert torch.equal(x16[..., :3].float().cpu(), x.permute(0, 2, 3, 1).to(torch.bfloat16).float()) and float(x16[..., 3:].abs().max()) == 0
    ref = F.conv2d(x.to(torch.bfloat16).double(), w.to(torch.bfloat16).double(), None, 2, 3)
    y = torch.full((N, ref.shape[2], ref.shape[3], 64), float('nan'), device=DEV, dtype=torch.bfloat16)
    igemm_t(x16, pack_t(w, 0, 2, 1, 16), None, y, 7, 2, 3, 0)
    report('stem conv bf16', nchw(y.float()), ref, 2.0**-9 * float(ref.abs().max()), 2.0**-8)
    # its weight gradient: X has 16 stored channels, 3 real ones
    dy = rnd(N, 64, ref.shape[2], ref.shape[3], seed=3)
    dys = nhwc(dy).to(torch.bfloat16)
    slab = torch.empty(L().dbn_wgrad_slab_floats(N, ref.shape[2], ref.shape[3], 64, 16, 7, 7), device=DEV)
    g = torch.full((64, 3, 7, 7), float('nan'), device=DEV)
    _lib.check(L().dbn_wgrad_t(1, 1, dys.data_ptr(), x16.data_ptr(), slab.data_ptr(), g.data_ptr(), N, ref.shape[2], ref.shape[3], 64, H, W,
                               16, 3, 7, 7, 2, 3, 1.0, stream()), 'wgrad_t')
    xr = x.to(torch.bfloat16).double().requires_grad_(False)
    wr = w.double().requires_grad_(True)
    (gref, ) = torch.autograd.grad(F.conv2d(xr, wr, None, 2, 3), wr, dy.to(torch.bfloat16).double())
    report('stem wgrad bf16', g.cpu(), gref, 1e-5 * float(gref.abs().max()), 1e-4)


@pytest.mark.parametrize('case', [(2, 64, 64, 3, 1, 1, 20, 12), (3, 128, 256, 1, 1, 0, 9, 7), (2, 64, 128, 3, 2, 1, 16, 16), (1, 256, 256, 3, 1, 1, 12, 12),
                                  (2, 64, 64, 2, 2, 0, 10, 12)])
@pytest.mark.parametrize('variant', [0, 2])
def test_weight_gradient_on_bf16_storage(case, variant):
    """dbn_wgrad_t(at = bf16): dY and X are read as stored and multiplied on the bf16 matrix pipe with fp32 accumulation into fp32
    slabs: equals the fp64 weight gradient of the ROUNDED tensors to fp32 accumulation noise.  variant 0 (default): LDS-DMA
    panels in memory order + transposing LDS reads (wgrad_tr_kernel); variant 2: the register-transposing kernel."""
    N, Ci, Co, k, s_, p_, H, W = case
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, k, k, seed=2).double().requires_grad_(True)
    y = F.conv2d(x.to(torch.bfloat16).double(), w, None, s_, p_)
    dy = rnd(*y.shape, seed=3)
    (gref, ) = torch.autograd.grad(y, w, dy.to(torch.bfloat16).double())
    xs, dys = nhwc(x).to(torch.bfloat16), nhwc(dy).to(torch.bfloat16)
    Ho, Wo = y.shape[2], y.shape[3]
    slab = torch.empty(L().dbn_wgrad_slab_floats(N, Ho, Wo, Co, Ci, k, k), device=DEV)
    g = torch.full((Co, Ci, k, k), float('nan'), device=DEV)
    try:
        _lib.check(L().dbn_set_wgrad_variant(variant), 'variant')
        _lib.check(L().dbn_wgrad_t(1, 1, dys.data_ptr(), xs.data_ptr(), slab.data_ptr(), g.data_ptr(), N, Ho, Wo, Co, H, W, Ci, Ci, k, k, s_, p_,
                                   0.5, stream()), 'wgrad_t')
        torch.cuda.synchronize()
    finally:
        L().dbn_set_wgrad_variant(0)
    report('wgrad bf16 storage variant %d' % variant, g.cpu(), 0.5 * gref, 2e-5 * float(gref.abs().max()), 1e-4)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_batchnorm_and_pool_kernels_on_16bit_storage(dtype):
    """The HBM-bound kernels with 16-bit tensors == the fp32 kernels on the same (rounded) inputs, up to one output rounding."""
    at = AT_OF[dtype]
    eps = 2.0**-8 if dtype == torch.bfloat16 else 2.0**-10
    N, C, H, W = 3, 64, 10, 14
    M = N * H * W
    y32 = nhwc(rnd(N, C, H, W, seed=1)).to(dtype).float()
    res32 = nhwc(rnd(N, C, H, W, seed=2)).to(dtype).float()
    dout32 = nhwc(rnd(N, C, H, W, seed=3)).to(dtype).float()
    gamma, beta = (rnd(C, seed=4) * 0.3 + 1).to(DEV), rnd(C, seed=5).to(DEV)
    ws = reduce_ws()

    def run(at_, conv):
        y, res, dout = conv(y32), conv(res32), conv(dout32)
        dt = y.dtype
        sc, sh, mu, rs = (torch.empty(C, device=DEV) for _ in range(4))
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        _lib.check(L().dbn_bn_train_stats_t(at_, y.data_ptr(), M, C, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, rm.data_ptr(),
                                            rv.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), ws.data_ptr(),
                                            stream()), 'stats')
        out = torch.full(y.shape, float('nan'), device=DEV, dtype=dt)
        _lib.check(L().dbn_bn_apply_t(at_, y.data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr(), None, None, out.data_ptr(), M, C, 1,
                                      stream()), 'apply')
        dy = torch.full(y.shape, float('nan'), device=DEV, dtype=dt)
        gout = torch.full(y.shape, float('nan'), device=DEV, dtype=dt)
        dg, db, dbias = (torch.empty(C, device=DEV) for _ in range(3))
        _lib.check(L().dbn_bn_backward_t(at_, None, 0, y.data_ptr(), out.data_ptr(), None, None, dout.data_ptr(), mu.data_ptr(), rs.data_ptr(),
                                         gamma.data_ptr(), dy.data_ptr(), gout.data_ptr(), 0, dg.data_ptr(), db.data_ptr(), dbias.data_ptr(),
                                         M, C, 1.0, ws.data_ptr(), stream()), 'bn backward')
        pool = torch.full((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), float('nan'), device=DEV, dtype=dt)
        _lib.check(L().dbn_bnrelu_maxpool_fwd_t(at_, y.data_ptr(), sc.data_ptr(), sh.data_ptr(), pool.data_ptr(), N, H, W, C, stream()), 'pool')
        dpool = conv(nhwc(rnd(N, C, pool.shape[1], pool.shape[2], seed=7)).to(dtype).float())
        dz = torch.full(y.shape, float('nan'), device=DEV, dtype=dt)
        nparts = L().dbn_maxpool_bwd_parts(N, H, W, C)
        parts = torch.empty(2 * C * nparts, device=DEV)
        _lib.check(L().dbn_bnrelu_maxpool_bwd_t(at_, y.data_ptr(), sc.data_ptr(), sh.data_ptr(), pool.data_ptr(), dpool.data_ptr(), dz.data_ptr(),
                                                N, H, W, C, mu.data_ptr(), rs.data_ptr(), parts.data_ptr(), stream()), 'pool bwd')
        # the BatchNorm backward fed with those partial sums == the one that reduces dz itself
        dy_a, dy_b = (torch.full(y.shape, float('nan'), device=DEV, dtype=dt) for _ in range(2))
        dg_a, db_a, dg_b, db_b = (torch.empty(C, device=DEV) for _ in range(4))
        for sums, np_, dyo, dgo, dbo in ((parts, nparts, dy_a, dg_a, db_a), (None, 0, dy_b, dg_b, db_b)):
            _lib.check(L().dbn_bn_backward_t(at_, None if sums is None else sums.data_ptr(), np_, y.data_ptr(), None, None, None, dz.data_ptr(),
                                             mu.data_ptr(), rs.data_ptr(), gamma.data_ptr(), dyo.data_ptr(), None, 0, dgo.data_ptr(),
                                             dbo.data_ptr(), None, M, C, 1.0, ws.data_ptr(), stream()), 'bn backward from pool sums')
        report('pool-fused BN sums: dgamma', dg_a.cpu(), dg_b.cpu(), 2e-5 * float(dg_b.abs().max()), 1e-4)
        report('pool-fused BN sums: dbeta', db_a.cpu(), db_b.cpu(), 2e-5 * float(db_b.abs().max()), 1e-4)
        report('pool-fused BN sums: dy', dy_a.float().cpu(), dy_b.float().cpu(), 2.0**-7 * float(dy_b.float().abs().max()), 2.0**-7)
        up = torch.full((N, 2 * H, 2 * W, C), float('nan'), device=DEV, dtype=dt)
        _lib.check(L().dbn_nearest_up_fwd_t(at_, y.data_ptr(), None, up.data_ptr(), N, H, W, C, 2 * H, 2 * W, C, 0, stream()), 'up')
        dn = torch.full(y.shape, float('nan'), device=DEV, dtype=dt)
        _lib.check(L().dbn_nearest_up_bwd_t(at_, up.data_ptr(), dn.data_ptr(), N, H, W, C, 2 * H, 2 * W, C, 0, 0, stream()), 'up bwd')
        return {k: v.float().cpu() for k, v in dict(sc=sc, sh=sh, out=out, dy=dy, gout=gout, dg=dg, db=db, dbias=dbias, pool=pool, dz=dz,
                                                     up=up, dn=dn).items()}

    a = run(at, lambda t: t.to(dtype))
    b = run(0, lambda t: t.clone())
    for k in a:
        tol = 1e-5 if k in ('sc', 'sh', 'dg', 'db') else eps
        if k == 'dbias':  # column sums of dy: of the ROUNDED dy in 16-bit storage... of values that cancel analytically
            continue
        if k == 'dz':  # equality with the pooled maximum is tested on rounded values: ties can differ from the fp32 run
            assert float((a[k] - b[k]).abs().gt(eps * (1 + b[k].abs())).float().mean()) < 0.02
            continue
        report('16-bit %s %s' % (dtype, k), a[k], b[k], tol * float(b[k].abs().max()), tol)


needs_experiments = pytest.mark.skipif(not _lib.lib().dbn_has_experiments(),
                                       reason='variant measured slower and left out of the product library (make -C csrc EXP=1 builds it)')


@needs_experiments
@pytest.mark.parametrize('tile', [0, 1, 2, 4])
def test_presplit_bf16x3_operands(tile):
    """dbn_split3 + the at = DBN_AT_SPLIT3 entry points: an fp32 tensor split once into three bf16 planes (a0 + a1 + a2 == a
    exactly) and gathered as stored gives BIT-IDENTICAL results to the ns = 3 path that splits at staging time (same terms, same
    MFMA order) — forward conv, data gradient, strided transposed conv, weight gradient."""
    N, Ci, Co, H, W = 3, 64, 128, 14, 18
    x = nhwc(rnd(N, Ci, H, W, seed=1))
    planes = torch.empty((3, ) + tuple(x.shape), device=DEV, dtype=torch.bfloat16)
    _lib.check(L().dbn_split3(x.data_ptr(), planes.data_ptr(), x.numel(), stream()), 'split3')
    assert torch.equal(planes.float().sum(0), x), 'the three bf16 terms do not sum to the fp32 value exactly'
    assert float((planes[1].float().abs() - 2.0**-7 * planes[0].float().abs()).clamp_min(0).max()) == 0  # |a1| <= ulp(a0)/2
    for (k, s_, p_, mode, wshape) in ((3, 1, 1, 0, (Co, Ci, 3, 3)), (3, 2, 1, 0, (Co, Ci, 3, 3)), (3, 1, 1, 1, (Ci, Co, 3, 3)), (2, 2, 0, 1, (Ci, Co, 2, 2))):
        w = rnd(*wshape, seed=2, scale=0.05)
        wpk = pack_t(w, mode, s_, 3, Ci if mode == 0 else 0)
        Hd = (H + 2 * p_ - k) // s_ + 1 if mode == 0 else (H if s_ == 1 else 2 * H)
        Wd = (W + 2 * p_ - k) // s_ + 1 if mode == 0 else (W if s_ == 1 else 2 * W)
        ya, yb = (torch.full((N, Hd, Wd, Co), float('nan'), device=DEV) for _ in range(2))
        igemm_t(x, wpk, None, ya, k, s_, p_, mode, tile=tile, ns=3)
        _lib.check(L().dbn_igemm_t(3, 3, planes.data_ptr(), wpk.data_ptr(), None, yb.data_ptr(), N, H, W, Ci, Hd, Wd, Co, k, k, s_, p_, mode,
                                   0, tile, 1, None, stream()), 'igemm_t split3')
        assert torch.equal(ya, yb), ('pre-split operands change the result', k, s_, mode)
    dy = nhwc(rnd(N, Co, H, W, seed=5))
    dyp = torch.empty((3, ) + tuple(dy.shape), device=DEV, dtype=torch.bfloat16)
    _lib.check(L().dbn_split3(dy.data_ptr(), dyp.data_ptr(), dy.numel(), stream()), 'split3')
    slab = torch.empty(L().dbn_wgrad_slab_floats_hw(N, H, W, Co, H, W, Ci, 3, 3, 4), device=DEV)
    ga, gb = (torch.full((Co, Ci, 3, 3), float('nan'), device=DEV) for _ in range(2))
    _lib.check(L().dbn_wgrad_t(0, 3, dy.data_ptr(), x.data_ptr(), slab.data_ptr(), ga.data_ptr(), N, H, W, Co, H, W, Ci, Ci, 3, 3, 1, 1, 1.0,
                               stream()), 'wgrad ns3')
    _lib.check(L().dbn_wgrad_t(3, 3, dyp.data_ptr(), planes.data_ptr(), slab.data_ptr(), gb.data_ptr(), N, H, W, Co, H, W, Ci, Ci, 3, 3, 1, 1,
                               1.0, stream()), 'wgrad split3')
    assert torch.equal(ga, gb)


@needs_experiments
@pytest.mark.parametrize('case', [(2, 64, 64, 3, 1, 1, 20, 12), (3, 128, 256, 1, 1, 0, 9, 7), (2, 64, 128, 3, 2, 1, 16, 16), (1, 256, 256, 3, 1, 1, 12, 12),
                                  (2, 3, 64, 7, 2, 3, 32, 40), (2, 64, 64, 2, 2, 0, 10, 12)])
def test_weight_gradient_lds_dma_variant(case):
    """dbn_set_wgrad_variant(1): the LDS-DMA weight-gradient kernel (operands staged pixel-major by buffer_load ... lds, out-of-range
    lanes delivering zeros; no register transposes; natural slab order) == autograd of F.conv2d, like the default kernel."""
    N, Ci, Co, k, s_, p_, H, W = case
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, k, k, seed=2).requires_grad_(True)
    y = F.conv2d(x, w, None, s_, p_)
    dy = rnd(*y.shape, seed=3)
    (gref, ) = torch.autograd.grad(y, w, dy)
    xs = nhwc(pad_c(x, (Ci + 3) // 4 * 4))
    try:
        L().dbn_set_wgrad_variant(1)
        g = wgrad(nhwc(dy), xs, Co, Ci, k, s_, p_, scale=0.5)
    finally:
        L().dbn_set_wgrad_variant(0)
    g0 = wgrad(nhwc(dy), xs, Co, Ci, k, s_, p_, scale=0.5)
    report('wgrad (LDS-DMA variant)', g.cpu(), 0.5 * gref, 2e-5 * float(gref.abs().max()), 1e-4)
    report('wgrad variants agree', g.cpu(), g0.cpu(), 1e-5 * float(gref.abs().max()), 1e-5)


@pytest.mark.parametrize('mode_name', ['f32', 'bf16x3', 'bf16c', 'bf16', 'fp16'])
@pytest.mark.parametrize('Co,tile', [(64, 3), (128, 1), (128, 3), (64, 0), (128, 0)])
def test_pixel_patch_convolution_equals_the_gather_form(mode_name, Co, tile):
    """3x3 / stride-1 convolutions and their data gradients run in the pixel-patch form (the input patch of an 8 x 16 output tile
    staged in LDS once per channel block — 32 channels in the 16-bit matrix modes, 16 in exact fp32 (round 4) — taps as LDS
    offsets).  Same k order, same MFMA sequence per accumulator as the generic gather loop: the results are BIT-IDENTICAL to it
    (dbn_set_patch_conv(0)), including the fused bias, the accumulate form and, through it, everything the generic loop is tested
    against.  tile 0: the library's own choice (exact fp32: the 128 x 64 patch kernel against the 64 x 64 gather tile)."""
    ns, dtype = {'f32': (0, torch.float32), 'bf16x3': (3, torch.float32), 'bf16c': (1, torch.float32), 'bf16': (1, torch.bfloat16),
                 'fp16': (1, torch.float16)}[mode_name]
    kind = 2 if dtype == torch.float16 else ns
    N, Ci, H, W = 2, 128, 24, 48   # four 32-channel blocks, 3 x 3 patches per image
    x = nhwc(rnd(N, Ci, H, W, seed=11)).to(dtype)
    w = rnd(Co, Ci, 3, 3, seed=12, scale=(2.0 / (Ci * 9))**0.5)
    b = rnd(Co, seed=13).to(DEV)
    dy = nhwc(rnd(N, Co, H, W, seed=14)).to(dtype)
    base = nhwc(rnd(N, Ci, H, W, seed=15)).to(dtype)

    def run():
        y = torch.full((N, H, W, Co), float('nan'), device=DEV, dtype=dtype)
        igemm_t(x, pack_t(w, 0, 1, kind, Ci), b, y, 3, 1, 1, 0, ns=ns, tile=tile)  # (the patch form needs a 128-row tile: 1 or 3)
        d = base.clone()
        igemm_t(dy, pack_t(w, 1, 1, kind), None, d, 3, 1, 1, 1, accumulate=1, ns=ns, tile=tile)
        torch.cuda.synchronize()
        return y, d

    try:
        L().dbn_set_wres16(0)  # (round 6: the weight-resident kernel takes 128 -> 128 on 16-bit storage — another summation order; tests/test_wres16_gpu.py)
        assert L().dbn_set_patch_conv(3 if ns == 0 else 1) in (0, 1, 2, 3)  # (3: exact fp32 takes the patch kernel wherever it is eligible)
        if tile in (0, 3):  # (exact fp32 has the 128 x 64 patch kernel only; 128 x 128 stays on the gather loop)
            cfg = L().dbn_igemm_kernel_config(AT_OF[dtype], ns, 0, N, H, W, Ci, H, W, Co, 3, 3, 1, 1, tile, 1)
            assert cfg & 16, 'the patch kernel was not selected (config %d)' % cfg
        y1, d1 = run()
        L().dbn_set_patch_conv(0)
        y0, d0 = run()
    finally:
        L().dbn_set_patch_conv(1)
        L().dbn_set_wres16(1)
    assert torch.isfinite(y1.float()).all() and torch.isfinite(d1.float()).all()
    assert torch.equal(y1, y0), 'forward: max |diff| %g' % float((y1.float() - y0.float()).abs().max())
    assert torch.equal(d1, d0), 'data gradient: max |diff| %g' % float((d1.float() - d0.float()).abs().max())
    # and against fp64: on the operands as stored for 16-bit storage; bf16x3 is fp32-accurate on fp32 operands
    wq = w.to(dtype).double() if dtype != torch.float32 else w.double()
    ref = F.conv2d(nchw(x.double()), wq, b.double().cpu(), 1, 1)
    tol = {'f32': 2e-6, 'bf16x3': 2e-6, 'bf16c': 2e-2, 'bf16': 2e-2, 'fp16': 3e-3}[mode_name]
    err = float((nchw(y1.double()) - ref).abs().max()) / float(ref.abs().max())
    assert err < tol, err


@pytest.mark.parametrize('mode_name', ['f32', 'bf16', 'bf16x3'])
def test_pixel_patch_convolution_is_race_free_at_full_size(mode_name):
    """The pixel-patch kernels pipeline LDS-DMA weight stages and register-staged patches across raw barriers (counted vmcnt,
    lgkmcnt(0) before each barrier).  A missing wait does not show on small grids — every workgroup alone on its CU — but did at
    the benchmark's size (several workgroups per CU, thousands of tiles): repeat full-size launches (16 x 160 x 160, both 128-row
    tiles, forward and data gradient) and require every run to equal the generic gather form bit for bit."""
    ns, dtype = {'f32': (0, torch.float32), 'bf16x3': (3, torch.float32), 'bf16': (1, torch.bfloat16)}[mode_name]
    N, H, W = 16, 160, 160
    g = torch.Generator(device=DEV).manual_seed(3)
    L().dbn_set_wres16(0)  # (this test is about the pixel-patch kernel; the weight-resident one has its own: tests/test_wres16_gpu.py)
    for (Ci, Co, tile, mode) in ((64, 256, 1, 1), (256, 64, 3, 0), (128, 128, 1, 1), (128, 128, 1, 0)):
        if ns == 0:
            tile = 3  # (the exact-fp32 patch kernel exists for the 128 x 64 tile)
        w = rnd(Co, Ci, 3, 3, seed=2, scale=0.05)
        cin, cout = (Ci, Co) if mode == 0 else (Co, Ci)
        x = torch.randn(N, H, W, cin, device=DEV, generator=g).to(dtype)
        wp = pack_t(w, mode, 1, ns, Ci) if mode == 0 else pack_t(w, 1, 1, ns)
        try:
            L().dbn_set_patch_conv(0)
            ref = torch.zeros(N, H, W, cout, device=DEV, dtype=dtype)
            igemm_t(x, wp, None, ref, 3, 1, 1, mode, ns=ns, tile=tile)
        finally:
            L().dbn_set_patch_conv(1)
        for run in range(3):
            y = torch.zeros_like(ref)
            igemm_t(x, wp, None, y, 3, 1, 1, mode, ns=ns, tile=tile)
            torch.cuda.synchronize()
            assert torch.equal(y, ref), '%s Ci %d Co %d tile %d mode %d run %d: %d elements differ' % (
                mode_name, Ci, Co, tile, mode, run, int((y != ref).sum()))
    L().dbn_set_wres16(1)


def test_image_chunking_on_bf16_storage():
    """The same chunking (dbn_set_index_limits) on stored-bf16 tensors: conv + fused BN statistics (pixel-patch kernel: H % 8 = 0,
    W % 16 = 0), a stride-2 conv (LDS-DMA ring), the stride-1 data gradient and the weight gradient (wgrad_tr_kernel) — chunked ==
    single launch bit for bit (the weight gradient to fp32 summation-order noise)."""
    bf = torch.bfloat16
    N, Ci, Co, H, W = 5, 64, 128, 24, 16
    x = nhwc(rnd(N, Ci, H, W, seed=1)).to(bf)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=0.05)
    bias = rnd(Co, seed=4).to(DEV)
    dy = nhwc(rnd(N, Co, H, W, seed=5)).to(bf)

    def run_all():
        out = {}
        y = torch.zeros((N, H, W, Co), device=DEV, dtype=bf)
        g_, b_ = torch.ones(Co, device=DEV), torch.zeros(Co, device=DEV)
        rm_, rv_ = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
        sc, sh, mu, rs = (torch.empty(Co, device=DEV) for _ in range(4))
        ws = torch.empty(L().dbn_conv_bn_ws_floats(N, H, W, Co, 0, 1), device=DEV)
        _lib.check(L().dbn_conv_bn_t(1, x.data_ptr(), pack_t(w, 0, 1, 1, Ci).data_ptr(), bias.data_ptr(), y.data_ptr(), N, H, W, Ci, H, W, Co,
                                     3, 3, 1, 1, 0, 0, 3, 1, g_.data_ptr(), b_.data_ptr(), 1e-5, 0.1, rm_.data_ptr(), rv_.data_ptr(),
                                     sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), ws.data_ptr(), stream()), 'conv_bn_t')
        out['conv'], out['scale'], out['run_var'] = y, sc, rv_
        y2 = torch.zeros((N, H // 2, W // 2, Co), device=DEV, dtype=bf)
        igemm_t(x, pack_t(w, 0, 2, 1, Ci), bias, y2, 3, 2, 1, 0)
        out['conv_s2'] = y2
        d = torch.zeros((N, H, W, Ci), device=DEV, dtype=bf)
        igemm_t(dy, pack_t(w, 1, 1, 1), None, d, 3, 1, 1, 1, tile=3)
        out['dgrad'] = d
        slab = torch.empty(L().dbn_wgrad_slab_floats_hw(N, H, W, Co, H, W, Ci, 3, 3, 2), device=DEV)
        g = torch.zeros((Co, Ci, 3, 3), device=DEV)
        _lib.check(L().dbn_wgrad_t(1, 1, dy.data_ptr(), x.data_ptr(), slab.data_ptr(), g.data_ptr(), N, H, W, Co, H, W, Ci, Ci, 3, 3, 1, 1,
                                   1.0, stream()), 'wgrad_t')
        out['wgrad'] = g
        torch.cuda.synchronize()
        return out

    ref = run_all()
    try:
        L().dbn_set_index_limits(2 * H * W + 7, 0, 0)   # at most two images per launch
        got = run_all()
        L().dbn_set_index_limits(0, H * W * Co * 2 + 64, 0)  # one image of the widest (bf16) tensor per launch
        got1 = run_all()
    finally:
        L().dbn_set_index_limits(0, 0, 0)
    for k, v in ref.items():
        for tag, gk in (('2/launch', got[k]), ('1/launch', got1[k])):
            if k == 'wgrad':
                report('chunked bf16 wgrad ' + tag, gk.cpu(), v.cpu(), 1e-5 * float(v.abs().max()), 1e-5)
            elif k in ('scale', 'run_var'):
                report('chunked bf16 %s %s' % (k, tag), gk.cpu(), v.cpu(), 1e-7, 1e-6)
            else:
                assert torch.equal(gk, v), (k, tag)


def test_dma_ring_kernels_are_race_free_at_full_size():
    """The generic 16-bit convolution loop (LDS-DMA ring; here a stride-2 3x3 conv and a parity-class data gradient) and the bf16
    weight gradient (LDS-DMA + transposing reads) at the benchmark's size (16 x 160 x 160): every repetition is bit-identical to
    the first, and the results agree with an independent evaluation (torch on the same rounded operands / the register-transposing
    weight-gradient kernel)."""
    bf = torch.bfloat16
    N, H, W, Ci, Co = 16, 160, 160, 64, 128
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn(N, H, W, Ci, device=DEV, generator=g).to(bf)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=0.05)
    wq = w.to(bf).float().to(DEV)
    # stride-2 forward conv
    wp = pack_t(w, 0, 2, 1, Ci)
    outs = []
    for _ in range(3):
        y = torch.zeros(N, H // 2, W // 2, Co, device=DEV, dtype=bf)
        igemm_t(x, wp, None, y, 3, 2, 1, 0)
        torch.cuda.synchronize()
        outs.append(y)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), 'stride-2 conv differs between runs'
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wq, None, 2, 1).permute(0, 2, 3, 1)
    assert float((outs[0].float() - ref).abs().max()) <= 2.0**-7 * float(ref.abs().max())
    # its data gradient (four parity classes)
    dy = torch.randn(N, H // 2, W // 2, Co, device=DEV, generator=g).to(bf)
    wpd = pack_t(w, 1, 2, 1)
    outs = []
    for _ in range(3):
        d = torch.zeros(N, H, W, Ci, device=DEV, dtype=bf)
        igemm_t(dy, wpd, None, d, 3, 2, 1, 1)
        torch.cuda.synchronize()
        outs.append(d)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), 'stride-2 data gradient differs between runs'
    ref = F.conv_transpose2d(dy.float().permute(0, 3, 1, 2), wq, None, 2, 1, output_padding=1).permute(0, 2, 3, 1)
    assert float((outs[0].float() - ref).abs().max()) <= 2.0**-7 * float(ref.abs().max())
    # weight gradient of a 3x3 / stride-1 layer: wgrad_tr_kernel, repeated, against the register-transposing kernel
    xs = torch.randn(N, H, W, Ci, device=DEV, generator=g).to(bf)
    dys = torch.randn(N, H, W, Ci, device=DEV, generator=g).to(bf)
    slab = torch.empty(L().dbn_wgrad_slab_floats_hw(N, H, W, Ci, H, W, Ci, 3, 3, 2), device=DEV)

    def wg(variant):
        gr = torch.zeros(Ci, Ci, 3, 3, device=DEV)
        try:
            _lib.check(L().dbn_set_wgrad_variant(variant), 'variant')
            _lib.check(L().dbn_wgrad_t(1, 1, dys.data_ptr(), xs.data_ptr(), slab.data_ptr(), gr.data_ptr(), N, H, W, Ci, H, W, Ci, Ci, 3, 3,
                                       1, 1, 1.0, stream()), 'wgrad_t')
            torch.cuda.synchronize()
        finally:
            L().dbn_set_wgrad_variant(0)
        return gr

    runs = [wg(0) for _ in range(3)]
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2]), 'bf16 weight gradient differs between runs'
    other = wg(2)
    report('wgrad_tr vs register-transposing kernel', runs[0].cpu(), other.cpu(), 2e-5 * float(other.abs().max()), 1e-4)


@pytest.mark.parametrize('mode_name', ['bf16x3', 'bf16c', 'bf16'])
@pytest.mark.parametrize('case', [(3, 64, 64, 12, 32), (2, 128, 64, 8, 16), (2, 64, 128, 16, 48), (1, 256, 256, 4, 16), (5, 64, 192, 20, 16)])
def test_pixel_patch_weight_gradient(case, mode_name):
    """3x3 / stride-1 weight gradients in the 16-bit matrix modes (H % 4 = 0, W % 16 = 0, channels in blocks of 64) run in the
    pixel-patch form (wgrad_patch_kernel: dY patch and the X rows of a tap row staged in LDS once, taps as LDS offsets, transposing
    LDS reads).  Reference: the fp64 weight gradient on the operands as the mode sees them (stored bf16 / rounded to bf16 / exact
    for the three-way split); and the generic kernels (dbn_set_wgrad_variant(2)) must agree to summation-order noise."""
    N, Ci, Co, H, W = case
    ns, at = {'bf16x3': (3, 0), 'bf16c': (1, 0), 'bf16': (1, 1)}[mode_name]
    x = rnd(N, Ci, H, W, seed=1)
    dy = rnd(N, Co, H, W, seed=3)
    rq = (lambda t: t.to(torch.bfloat16).double()) if ns == 1 else (lambda t: t.double())
    w = rnd(Co, Ci, 3, 3, seed=2).double().requires_grad_(True)
    (gref, ) = torch.autograd.grad(F.conv2d(rq(x), w, None, 1, 1), w, rq(dy))
    dt = torch.bfloat16 if at == 1 else torch.float32
    xs, dys = nhwc(x).to(dt), nhwc(dy).to(dt)
    slab = torch.empty(L().dbn_wgrad_slab_floats_hw(N, H, W, Co, H, W, Ci, 3, 3, 2 if at == 1 else 4), device=DEV)

    def run(variant):
        g = torch.full((Co, Ci, 3, 3), float('nan'), device=DEV)
        try:
            _lib.check(L().dbn_set_wgrad_variant(variant), 'variant')
            _lib.check(L().dbn_wgrad_t(at, ns, dys.data_ptr(), xs.data_ptr(), slab.data_ptr(), g.data_ptr(), N, H, W, Co, H, W, Ci, Ci, 3, 3,
                                       1, 1, 0.5, stream()), 'wgrad_t')
            torch.cuda.synchronize()
        finally:
            L().dbn_set_wgrad_variant(0)
        return g.cpu()

    g = run(0)
    scale = float(gref.abs().max())
    report('patch wgrad %s %s' % (mode_name, case), g, 0.5 * gref, 2e-5 * scale, 1e-4)
    report('patch vs generic kernels', g, run(2), 2e-5 * scale, 1e-4)


# N, Cin (of the forward conv = channels of dx), Cout, k, s, p, H, W, as_forward
BNSUM_CASES = [(2, 128, 128, 3, 1, 1, 16, 32, False), (1, 64, 256, 3, 1, 1, 8, 64, False), (2, 256, 64, 3, 1, 1, 8, 80, True),  # (the weight-resident kernel, csrc/wres16.hip)
               (2, 64, 64, 3, 1, 1, 16, 12, False), (1, 256, 128, 3, 1, 1, 12, 12, False), (3, 128, 64, 1, 1, 0, 9, 7, False),
               (2, 64, 64, 2, 2, 0, 10, 12, True), (1, 64, 128, 3, 2, 1, 18, 14, True), (2, 64, 64, 3, 1, 1, 40, 24, False),
               (1, 64, 128, 3, 2, 1, 18, 14, False), (2, 128, 256, 3, 2, 1, 16, 16, False), (1, 64, 64, 3, 2, 1, 17, 13, False)]


@pytest.mark.parametrize('case', BNSUM_CASES)
@pytest.mark.parametrize('tile', [0, 1, 2, 3, 4])
@pytest.mark.parametrize('mask', ['self', 'tensor'])
@pytest.mark.parametrize('accumulate', [0, 1])
@pytest.mark.parametrize('math', ['f32', 'bf16x3', 'bf16'])
def test_data_gradient_with_batchnorm_backward_sums(case, tile, mask, accumulate, math):
    """dbn_igemm_bnsums_t: a data gradient (mode 1, any stride) or a forward-form conv (mode 0: the ConvTranspose2d data gradient
    of the head) whose epilogue also reduces the two per-channel sums of the BatchNorm backward that consumes its output (the
    conv -> BN -> ReLU chain of resnet.py:70-91 / basic.py:32-36 backwards) — and of a second BatchNorm over the same gradient and
    mask (projection shortcut, resnet.py:84-91).  dst must equal the plain call's bit for bit, and the folded partials must
    equal sum(g), sum(g * xhat) with g = dz * [mask > 0] evaluated in fp64 on the final dz AS STORED.  'bf16x3': fp32 tensors,
    split-bf16 matrix math; 'bf16': bf16 tensors (incl. the pixel-patch kernels of the 3x3 / stride-1 shapes)."""
    N, Ci, Co, k, s, p, H, W, as_forward = case
    ns, at = {'f32': (0, 0), 'bf16x3': (3, 0), 'bf16': (1, 1)}[math]
    kind = {'f32': 0, 'bf16x3': 3, 'bf16': 1}[math]
    tdt = torch.bfloat16 if at == 1 else torch.float32
    if tile == 1 and (Co if as_forward else Ci) % 128 != 0:
        pytest.skip('128-wide tile needs Cd % 128 == 0')
    if math != 'f32' and (accumulate == 1 and tile not in (0, 3)):
        pytest.skip('16-bit math: the tile sweep runs without accumulate')
    if as_forward:  # dst = conv(src): mode 0
        src = rnd(N, Ci, H, W, seed=1)
        w = rnd(Co, Ci, k, k, seed=2, scale=(2.0 / (Ci * k * k))**0.5)
        dzc = F.conv2d(src, w, None, s, p)
        mode, Cd, wsrc, stride_pack = 0, Co, w, 1
    else:  # dst = d(input) of conv: mode 1
        x = rnd(N, Ci, H, W, seed=1).requires_grad_(True)
        w = rnd(Co, Ci, k, k, seed=2, scale=(2.0 / (Ci * k * k))**0.5)
        yf = F.conv2d(x, w, None, s, p)
        src = rnd(*yf.shape, seed=4)
        (dzc, ) = torch.autograd.grad(yf, x, src)
        mode, Cd, wsrc, stride_pack = 1, Ci, w, s
    Hd, Wd = dzc.shape[2:]
    srcs = nhwc(src).to(tdt)
    O_, I_, R_, S_ = wsrc.shape
    wpk = torch.empty(L().dbn_igemm_panel_floats_t(kind, O_, I_, R_, S_, mode, stride_pack, srcs.shape[3] if mode == 0 else 0), device=DEV)
    _lib.check(L().dbn_pack_weights_t(kind, wsrc.to(DEV).data_ptr(), O_, I_, R_, S_, mode, stride_pack, srcs.shape[3] if mode == 0 else 0,
                                      wpk.data_ptr(), stream()), 'pack')
    y = rnd(N, Cd, Hd, Wd, seed=7)
    mean, rstd = rnd(Cd, seed=8, scale=0.2), rnd(Cd, seed=9).abs() + 0.5
    msc, msh = rnd(Cd, seed=10), rnd(Cd, seed=11, scale=0.3)
    z = rnd(N, Cd, Hd, Wd, seed=12)
    old = rnd(N, Cd, Hd, Wd, seed=13)
    ys, zs = nhwc(y).to(tdt), nhwc(z).to(tdt)
    y, z = nchw(ys.float()), nchw(zs.float())  # the values the kernel sees (rounded for bf16 storage)
    fresh = lambda: nhwc(old).to(tdt).clone() if accumulate else torch.full((N, Hd, Wd, Cd), float('nan'), device=DEV, dtype=tdt)
    dst, plain = fresh(), fresh()
    geo = (N, srcs.shape[1], srcs.shape[2], srcs.shape[3], Hd, Wd, Cd, k, k, s, p, mode)
    _lib.check(L().dbn_igemm_t(at, ns, srcs.data_ptr(), wpk.data_ptr(), None, plain.data_ptr(), *geo, accumulate, tile, 1, None, stream()),
               'igemm_t')
    rows = L().dbn_igemm_bn_rows(at, ns, *geo[:11], mode, tile)
    assert rows > 0
    part = torch.full((2, Cd, rows), float('nan'), device=DEV)
    # a second BatchNorm over the same dz and mask tensor (projection shortcut), only with a mask tensor
    two = mask == 'tensor'
    y2s = nhwc(rnd(N, Cd, Hd, Wd, seed=21)).to(tdt)
    y2 = nchw(y2s.float())
    mean2, rstd2 = rnd(Cd, seed=22, scale=0.3), rnd(Cd, seed=23).abs() + 0.4
    mean2_d, rstd2_d = mean2.to(DEV), rstd2.to(DEV)
    part2 = torch.full((2, Cd, rows), float('nan'), device=DEV)
    dev = lambda t: t.to(DEV)
    mean_d, rstd_d, msc_d, msh_d = dev(mean), dev(rstd), dev(msc), dev(msh)
    # the in-kernel finalize (dbn_bnb_final): per-channel results without a separate fold launch; counters must come back zero
    import ctypes
    cnt = torch.zeros(L().dbn_igemm_bn_final_counters(rows, Cd), device=DEV, dtype=torch.int32)
    grp = torch.full((L().dbn_igemm_bn_final_group_floats(rows, Cd), ), float('nan'), device=DEV)
    fo = [torch.full((n_, ), float('nan'), device=DEV) for n_ in (2 * Cd, Cd, Cd, 2 * Cd, Cd, Cd)]
    fin = _lib.BnbFinal(cnt.data_ptr(), grp.data_ptr(), fo[0].data_ptr(), fo[1].data_ptr(), fo[2].data_ptr(),
                        fo[3].data_ptr() if two else None, fo[4].data_ptr() if two else None, fo[5].data_ptr() if two else None, 0.5)
    for rep in range(2):  # twice: the second call relies on the counters the first one left behind
        if rep:
            dst.copy_(fresh())
        _lib.check(L().dbn_igemm_bnsums_t(at, ns, srcs.data_ptr(), wpk.data_ptr(), None, dst.data_ptr(), *geo, accumulate, tile, ys.data_ptr(),
                                          zs.data_ptr() if mask == 'tensor' else None, None if mask == 'tensor' else msc_d.data_ptr(),
                                          None if mask == 'tensor' else msh_d.data_ptr(), mean_d.data_ptr(), rstd_d.data_ptr(),
                                          part.data_ptr(), y2s.data_ptr() if two else None, mean2_d.data_ptr() if two else None,
                                          rstd2_d.data_ptr() if two else None, part2.data_ptr() if two else None,
                                          ctypes.byref(fin), stream()), 'igemm_bnsums')
        torch.cuda.synchronize()
        assert int(cnt.abs().sum()) == 0, 'the finalize left a counter behind'
    torch.cuda.synchronize()
    # (round 6: a plain 3x3 / stride-1 call on bf16 storage may take the weight-resident kernel where the call with the sums epilogue stays on
    # the pixel-patch kernel — 64 -> 64 and 128 -> 128, csrc/wres16.hip: two summation orders, equal to a rounding step of the storage type)
    wres = lambda bnb: bool(at in (1, 2) and k == 3 and s == 1 and p == 1 and L().dbn_wres16_would_run(at, mode, N, Hd, Wd, srcs.shape[3], Cd, bnb, int(two)))
    if wres(0) == wres(1):
        assert torch.equal(dst, plain), 'the sums epilogue changed the convolution result'
    else:
        d_ = (dst.float() - plain.float()).abs()
        assert bool((d_ <= 4e-5 * float(plain.float().abs().max()) + 2.0**-7 * 1.01 * plain.float().abs()).all()), float(d_.max())
    dz = nchw(dst.float()).double()
    m32 = torch.addcmul(msh.view(1, -1, 1, 1), y, msc.view(1, -1, 1, 1))
    m = z.double() if mask == 'tensor' else m32.double()
    if mask == 'self':  # the kernel evaluates fmaf(y, sc, sh) in fp32: elements within round-off of zero may flip
        assert bool((m32.abs() > 1e-6).all()), 'test data has a mask value at round-off level'
    g = dz * (m > 0)
    xhat = (y.double() - mean.double().view(1, -1, 1, 1)) * rstd.double().view(1, -1, 1, 1)
    s1, s2 = g.sum((0, 2, 3)), (g * xhat).sum((0, 2, 3))
    got = part.double().sum(2).cpu()
    sc = float(g.abs().sum((0, 2, 3)).max()) + 1e-9
    report('bn-backward sums %s tile %d %s acc %d %s' % (case, tile, mask, accumulate, math), got, torch.stack([s1, s2]), 2e-6 * sc, 1e-5)
    Mtot = N * Hd * Wd
    report('finalized c1, c2', fo[0].cpu().view(2, Cd), torch.stack([s1, s2]) / Mtot, 2e-6 * sc / Mtot, 1e-5)
    report('finalized dgamma', fo[1].cpu(), 0.5 * s2, 2e-6 * sc, 1e-5)
    report('finalized dbeta', fo[2].cpu(), 0.5 * s1, 2e-6 * sc, 1e-5)
    if two:
        xhat2 = (y2.double() - mean2.double().view(1, -1, 1, 1)) * rstd2.double().view(1, -1, 1, 1)
        s2b = (g * xhat2).sum((0, 2, 3))
        report('second BatchNorm sums', part2.double().sum(2).cpu(), torch.stack([s1, s2b]), 2e-6 * sc, 1e-5)
        report('second BatchNorm finalized', torch.cat([fo[3].cpu(), fo[4].cpu(), fo[5].cpu()]),
               torch.cat([s1 / Mtot, s2b / Mtot, 0.5 * s2b, 0.5 * s1]), 2e-6 * sc, 1e-5)


@pytest.mark.parametrize('C,N,H,W,parts', [(64, 2, 12, 10, 1), (64, 2, 12, 10, 37), (128, 1, 16, 16, 700), (64, 2, 24, 24, 6400), (256, 1, 8, 8, 513)])
def test_batchnorm_backward_from_given_partial_sums(C, N, H, W, parts):
    """dbn_bn_backward_t with `sums` = [2][C][parts] partials produced elsewhere (a data gradient's epilogue writes one row per
    output tile: thousands of rows at bs16 160x160): the finalize step folds them — a 32-lane team per channel for few rows, a
    256-thread block per channel from 512 rows on — and the result must equal the BatchNorm + ReLU backward of autograd."""
    x = (rnd(N, C, H, W, seed=1) * 2 + 3).requires_grad_(True)
    gamma = (rnd(C, seed=2) * 0.3 + 1).requires_grad_(True)
    beta = rnd(C, seed=3).requires_grad_(True)
    z = F.relu(F.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-5))
    dout = rnd(N, C, H, W, seed=7)
    dx_ref, dg_ref, db_ref = torch.autograd.grad(z, (x, gamma, beta), dout)
    M = N * H * W
    xd = x.detach()
    mean = xd.mean((0, 2, 3))
    rstd = 1.0 / torch.sqrt(xd.var((0, 2, 3), unbiased=False) + 1e-5)
    g = (dout * (z.detach() > 0)).double()
    xhat = ((xd - mean.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1)).double()
    s1, s2 = g.sum((0, 2, 3)), (g * xhat).sum((0, 2, 3))
    # split the two sums into `parts` random addends per channel (fp32, as a producing kernel would store them)
    gen = torch.Generator().manual_seed(5)
    wts = torch.rand(2, C, parts, generator=gen, dtype=torch.float64) + 0.1
    wts = wts / wts.sum(2, keepdim=True)
    part = (torch.stack([s1, s2]).unsqueeze(2) * wts).float()
    resid = torch.stack([s1, s2]) - part.double().sum(2)  # keep the total exact to fp32 round-off of one addend
    part[:, :, 0] += resid.float()
    xs, zs, douts = nhwc(xd), nhwc(z.detach()), nhwc(dout)
    dy = torch.empty_like(xs)
    dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    dev = lambda t: t.contiguous().to(DEV)
    part_d, mean_d, rstd_d, gam_d = dev(part), dev(mean), dev(rstd), dev(gamma.detach())
    ws = reduce_ws()
    _lib.check(L().dbn_bn_backward_t(0, part_d.data_ptr(), parts, xs.data_ptr(), zs.data_ptr(), None, None, douts.data_ptr(),
                                     mean_d.data_ptr(), rstd_d.data_ptr(), gam_d.data_ptr(), dy.data_ptr(), None, 0, dg.data_ptr(),
                                     db.data_ptr(), None, M, C, 1.0, ws.data_ptr(), stream()), 'bn backward from sums')
    report('bn bwd (given sums, %d parts) dx' % parts, nchw(dy), dx_ref, 2e-5, 1e-4)
    report('dgamma', dg.cpu(), dg_ref, 1e-4, 1e-4)
    report('dbeta', db.cpu(), db_ref, 1e-4, 1e-4)


@pytest.mark.parametrize('N,Ci,Cs,Co,H,W', [(2, 64, 64, 64, 16, 32), (1, 128, 128, 128, 8, 16), (2, 32, 32, 64, 24, 16), (1, 20, 32, 192, 8, 48),
                                            (3, 256, 256, 64, 16, 16), (1, 64, 64, 64, 40, 40), (2, 32, 32, 64, 14, 30), (1, 32, 32, 64, 13, 30),
                                            (2, 64, 64, 64, 20, 20), (1, 32, 32, 64, 25, 25), (2, 32, 32, 64, 8, 96), (1, 64, 64, 128, 50, 50)])
@pytest.mark.parametrize('with_bn', [False, True])
def test_winograd_conv3x3(N, Ci, Cs, Co, H, W, with_bn):
    """dbn_winograd_conv_bn_f32: 3x3 / stride 1 / pad 1 forward convolution through Winograd F(2x2, 3x3) in fp32 (the BasicBlock, FPN
    smooth and head convs of resnet.py:70-91, segmentation_body.py:55-61, segmentation_head.py:24-25) against F.conv2d in fp64, with
    the bias and — with_bn — the folded train-mode BatchNorm statistics (scale / shift / saved mean / rstd / running statistics)
    against F.batch_norm.  Small maps (40 x 40: layer3, 20 x 20: layer4, odd sizes) run in the consecutive-tile form, wide maps whose
    size is not a multiple of the 8 x 16 patch with masked ragged patches.  fp32 arithmetic with another summation order than the direct form: tolerance 2e-6 of the output scale
    per element (the direct kernels' own fp32 rounding is ~3e-7 at K = 2304).  Cs > Ci: the source tensor carries padding channels."""
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=(2.0 / (Ci * 9))**0.5)
    b = rnd(Co, seed=3) * 0.1
    xs = torch.zeros(N, H, W, Cs, device=DEV)
    xs[..., :Ci] = nhwc(x)
    up = torch.full((L().dbn_winograd_panel_floats(Co, Cs), ), float('nan'), device=DEV)
    _lib.check(L().dbn_winograd_pack(w.to(DEV).data_ptr(), Co, Ci, Cs, 0, up.data_ptr(), stream()), 'winograd pack')
    assert L().dbn_winograd_eligible(N, H, W, Cs, Co)
    y = torch.full((N, H, W, Co), float('nan'), device=DEV)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    bd = b.to(DEV)
    if not with_bn:
        _lib.check(L().dbn_winograd_conv_bn_f32(xs.data_ptr(), up.data_ptr(), bd.data_ptr(), y.data_ptr(), N, H, W, Cs, Co, None, None, 0.0, 0.0,
                                                None, None, None, None, None, None, None, stream()), 'winograd')
    else:
        gamma, beta = (rnd(Co, seed=4) * 0.3 + 1).to(DEV), rnd(Co, seed=5).to(DEV)
        rm, rv = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
        sc, sh, mu, rs = (torch.full((Co, ), float('nan'), device=DEV) for _ in range(4))
        ws = torch.full((L().dbn_winograd_ws_floats(N, H, W, Co), ), float('nan'), device=DEV)
        _lib.check(L().dbn_winograd_conv_bn_f32(xs.data_ptr(), up.data_ptr(), bd.data_ptr(), y.data_ptr(), N, H, W, Cs, Co, gamma.data_ptr(),
                                                beta.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                                mu.data_ptr(), rs.data_ptr(), ws.data_ptr(), stream()), 'winograd+bn')
        mean = ref.mean((0, 2, 3))
        var = ref.var((0, 2, 3), unbiased=False)
        report('winograd BN mean', mu.cpu(), mean, 1e-5, 1e-5)
        report('winograd BN rstd', rs.cpu(), 1.0 / torch.sqrt(var + 1e-5), 1e-5, 1e-4)
        report('winograd BN scale', sc.cpu(), gamma.cpu().double() / torch.sqrt(var + 1e-5), 1e-5, 1e-4)
        report('winograd BN shift', sh.cpu(), beta.cpu().double() - mean * gamma.cpu().double() / torch.sqrt(var + 1e-5), 1e-5, 1e-4)
        n = N * H * W
        report('winograd BN running mean', rm.cpu(), 0.1 * mean, 1e-6, 1e-5)
        report('winograd BN running var', rv.cpu(), 0.9 + 0.1 * var * n / (n - 1), 1e-6, 1e-4)
    scale = float(ref.abs().max())
    report('winograd conv3x3', nchw(y), ref, 2e-6 * scale, 2e-6)
    # bit-reproducible from run to run, and against the direct kernel to fp32 rounding
    y2 = torch.full_like(y, float('nan'))
    _lib.check(L().dbn_winograd_conv_bn_f32(xs.data_ptr(), up.data_ptr(), bd.data_ptr(), y2.data_ptr(), N, H, W, Cs, Co, None, None, 0.0, 0.0,
                                            None, None, None, None, None, None, None, stream()), 'winograd')
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    yd = torch.full_like(y, float('nan'))
    wfull = torch.zeros(Co, Cs, 3, 3)
    wfull[:, :Ci] = w
    igemm(xs, pack(wfull, 0), bd, yd, 3, 1, 1, 0)
    report('winograd vs the direct kernel', y.cpu(), yd.cpu(), 3e-6 * scale, 3e-6)


@pytest.mark.parametrize('N,Ci,Co,H,W', [(2, 64, 64, 16, 32), (1, 128, 64, 8, 16), (2, 64, 256, 8, 32), (1, 64, 64, 40, 40), (2, 64, 64, 13, 30),
                                         (3, 64, 64, 20, 20), (1, 64, 128, 25, 25)])
@pytest.mark.parametrize('mask,accumulate', [('self', 0), ('tensor', 1), ('none', 1)])
def test_winograd_dgrad_with_bn_sums(N, Ci, Co, H, W, mask, accumulate):
    """dbn_winograd_dgrad_bnsums_f32: the data gradient of a 3x3 / stride-1 / pad-1 conv (Ci -> Co) through the Winograd kernel with
    the rotated / transposed filter panel, [+ accumulate], and in its epilogue the two sums of the BatchNorm backward that consumes dx
    (mask recomputed from the BatchNorm's own output, or a mask tensor with a SECOND BatchNorm over the same gradient) plus the
    in-kernel finalize — against autograd in fp64 (conv -> BN -> ReLU chain of resnet.py:70-91; same contract as
    test_batchnorm_backward_sums_in_the_data_gradient for the implicit-GEMM kernels)."""
    import ctypes
    x = rnd(N, Ci, H, W, seed=1).requires_grad_(True)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=(2.0 / (Ci * 9))**0.5)
    yf = F.conv2d(x.double(), w.double(), None, 1, 1)
    dyt = rnd(N, Co, H, W, seed=4)
    (dx_ref, ) = torch.autograd.grad(yf, x, dyt.double())
    old = rnd(N, Ci, H, W, seed=13)
    if accumulate:
        dx_ref = dx_ref + old.double()
    dys = nhwc(dyt)
    up = torch.full((L().dbn_winograd_panel_floats(Ci, Co), ), float('nan'), device=DEV)
    _lib.check(L().dbn_winograd_pack(w.to(DEV).data_ptr(), Ci, Co, Co, 1, up.data_ptr(), stream()), 'winograd pack (dgrad)')
    dx = nhwc(old).clone() if accumulate else torch.full((N, H, W, Ci), float('nan'), device=DEV)
    # (values on a 2^-6 grid: fma(y, sc, sh) is then exact in fp32, so the recomputed ReLU mask cannot differ from the fp64 one by round-off)
    grid = lambda t: torch.round(t * 64) / 64
    ybn = grid(rnd(N, Ci, H, W, seed=7))
    mean, rstd = rnd(Ci, seed=8, scale=0.2), rnd(Ci, seed=9).abs() + 0.5
    msc, msh = grid(rnd(Ci, seed=10)), grid(rnd(Ci, seed=11, scale=0.3)) + 1.0 / 128
    z = rnd(N, Ci, H, W, seed=12)
    y2 = rnd(N, Ci, H, W, seed=21)
    mean2, rstd2 = rnd(Ci, seed=22, scale=0.3), rnd(Ci, seed=23).abs() + 0.4
    rows = L().dbn_winograd_rows(N, H, W)
    d = lambda t: t.contiguous().to(DEV)
    ys, zs, y2s = nhwc(ybn), nhwc(z), nhwc(y2)
    mean_d, rstd_d, msc_d, msh_d, mean2_d, rstd2_d = d(mean), d(rstd), d(msc), d(msh), d(mean2), d(rstd2)
    two = mask == 'tensor'
    part = torch.full((2, Ci, rows), float('nan'), device=DEV)
    part2 = torch.full((2, Ci, rows), float('nan'), device=DEV)
    cnt = torch.zeros(L().dbn_igemm_bn_final_counters(rows, Ci), device=DEV, dtype=torch.int32)
    grp = torch.full((L().dbn_igemm_bn_final_group_floats(rows, Ci), ), float('nan'), device=DEV)
    fo = [torch.full((n_, ), float('nan'), device=DEV) for n_ in (2 * Ci, Ci, Ci, 2 * Ci, Ci, Ci)]
    fin = _lib.BnbFinal(cnt.data_ptr(), grp.data_ptr(), fo[0].data_ptr(), fo[1].data_ptr(), fo[2].data_ptr(),
                        fo[3].data_ptr() if two else None, fo[4].data_ptr() if two else None, fo[5].data_ptr() if two else None, 0.5)
    for rep in range(2):  # twice: the second call relies on the counters the first one left behind
        if rep and accumulate:
            dx.copy_(nhwc(old))
        if mask == 'none':
            _lib.check(L().dbn_winograd_dgrad_bnsums_f32(dys.data_ptr(), up.data_ptr(), dx.data_ptr(), N, H, W, Co, Ci, accumulate, None, None,
                                                         None, None, None, None, None, None, None, None, None, None, stream()), 'winograd dgrad')
        else:
            _lib.check(L().dbn_winograd_dgrad_bnsums_f32(dys.data_ptr(), up.data_ptr(), dx.data_ptr(), N, H, W, Co, Ci, accumulate, ys.data_ptr(),
                                                         zs.data_ptr() if two else None, None if two else msc_d.data_ptr(),
                                                         None if two else msh_d.data_ptr(), mean_d.data_ptr(), rstd_d.data_ptr(), part.data_ptr(),
                                                         y2s.data_ptr() if two else None, mean2_d.data_ptr() if two else None,
                                                         rstd2_d.data_ptr() if two else None, part2.data_ptr() if two else None,
                                                         ctypes.byref(fin), stream()), 'winograd dgrad + sums')
        torch.cuda.synchronize()
        assert int(cnt.abs().sum()) == 0, 'the finalize left a counter behind'
    scale = float(dx_ref.abs().max())
    report('winograd dgrad', nchw(dx), dx_ref, 3e-6 * scale, 3e-6)
    if mask == 'none':
        return
    dz = nchw(dx).double()  # the sums are over the kernel's own final values
    m32 = torch.addcmul(msh.view(1, -1, 1, 1), ybn, msc.view(1, -1, 1, 1))
    m = z.double() if two else m32.double()
    g = dz * (m > 0)
    xhat = (ybn.double() - mean.double().view(1, -1, 1, 1)) * rstd.double().view(1, -1, 1, 1)
    s1, s2 = g.sum((0, 2, 3)), (g * xhat).sum((0, 2, 3))
    sc = float(g.abs().sum((0, 2, 3)).max()) + 1e-9
    report('winograd bn-backward sums', part.double().sum(2).cpu(), torch.stack([s1, s2]), 2e-6 * sc, 1e-5)
    Mtot = N * H * W
    report('finalized c1, c2', fo[0].cpu().view(2, Ci), torch.stack([s1, s2]) / Mtot, 2e-6 * sc / Mtot, 1e-5)
    report('finalized dgamma', fo[1].cpu(), 0.5 * s2, 2e-6 * sc, 1e-5)
    report('finalized dbeta', fo[2].cpu(), 0.5 * s1, 2e-6 * sc, 1e-5)
    if two:
        xhat2 = (y2.double() - mean2.double().view(1, -1, 1, 1)) * rstd2.double().view(1, -1, 1, 1)
        s2b = (g * xhat2).sum((0, 2, 3))
        report('second BatchNorm sums', part2.double().sum(2).cpu(), torch.stack([s1, s2b]), 2e-6 * sc, 1e-5)
        report('second BatchNorm finalized', torch.cat([fo[3].cpu(), fo[4].cpu(), fo[5].cpu()]),
               torch.cat([s1 / Mtot, s2b / Mtot, 0.5 * s2b, 0.5 * s1]), 2e-6 * sc, 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('N,Ci,Cs,Co,H,W', [(2, 64, 64, 64, 16, 32), (1, 64, 64, 128, 20, 20), (2, 128, 128, 64, 9, 29), (1, 40, 64, 64, 8, 16),
                                            (3, 256, 256, 64, 24, 40), (1, 128, 128, 128, 40, 40), (5, 64, 64, 64, 13, 30), (2, 64, 64, 64, 64, 64),
                                            # consecutive-tile form (small maps): layer4 / layer3 shapes, odd sizes, one partial group
                                            (3, 64, 64, 64, 20, 20), (2, 128, 128, 64, 25, 25), (4, 64, 64, 128, 13, 13), (1, 64, 64, 64, 10, 9), (2, 64, 64, 64, 40, 36)])
def test_winograd_weight_gradient(N, Ci, Cs, Co, H, W):
    """dbn_winograd_wgrad_f32: weight gradient of a 3x3 / stride 1 / pad 1 conv through Winograd F(2x2, 3x3) over the tiles in fp32
    (torch.autograd's conv weight gradient of resnet.py:70-91, segmentation_body.py:55-61, segmentation_head.py:24-25) against
    autograd in fp64: ragged maps (zero-filled patch edges), odd sizes, padding channels (Cs > Ci), several channel blocks, scale.
    Tolerance 4e-6 of the gradient's scale per element (the direct kernel's own fp32 rounding is ~1e-6 here); bit-reproducible
    from run to run; against the direct kernel to fp32 rounding."""
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, 3, 3, seed=2).double().requires_grad_(True)
    y = F.conv2d(x.double(), w, None, 1, 1)
    dy = rnd(*y.shape, seed=4)
    (ref, ) = torch.autograd.grad(y, w, dy.double())
    xs, dys = nhwc(pad_c(x, Cs)), nhwc(dy)
    assert L().dbn_winograd_wgrad_eligible(N, H, W, Co, Cs, Ci)
    slab = torch.full((L().dbn_winograd_wgrad_slab_floats(N, H, W, Co, Cs), ), float('nan'), device=DEV)
    g = torch.full((Co, Ci, 3, 3), float('nan'), device=DEV)
    args = (dys.data_ptr(), xs.data_ptr(), None, None, slab.data_ptr(), g.data_ptr(), N, H, W, Co, Cs, Ci)
    _lib.check(L().dbn_winograd_wgrad_f32(3, *args, 0.5, stream()), 'winograd wgrad')
    scale = float(ref.abs().max())
    report('winograd wgrad', g.cpu(), 0.5 * ref, 4e-6 * 0.5 * scale, 4e-6)
    g2 = torch.full_like(g, float('nan'))
    slab.fill_(float('nan'))
    _lib.check(L().dbn_winograd_wgrad_f32(1, *args[:5], g2.data_ptr(), *args[6:], 0.5, stream()), 'winograd wgrad phase 1')
    _lib.check(L().dbn_winograd_wgrad_f32(2, *args[:5], g2.data_ptr(), *args[6:], 0.5, stream()), 'winograd wgrad phase 2')
    torch.cuda.synchronize()
    assert torch.equal(g, g2)
    gd = wgrad(dys, xs, Co, Ci, 3, 1, 1, scale=0.5)
    report('winograd wgrad vs the direct kernel', g.cpu(), gd.cpu(), 5e-6 * 0.5 * scale, 5e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('N,C,Co,H,W', [(2, 64, 64, 16, 32), (1, 128, 64, 20, 20), (2, 64, 128, 13, 30), (1, 256, 64, 40, 40), (3, 64, 64, 25, 25)])
def test_winograd_apply_on_load_equals_bn_apply_then_conv(N, C, Co, H, W):
    """dbn_winograd_conv_bn_act_f32 / dbn_winograd_wgrad_f32(x_scale, x_shift): the BatchNorm + ReLU in front of the conv applied while
    the kernels stage their patches (basic.py:32-36) against dbn_bn_apply followed by the same kernels on the written activation —
    bit for bit (the same fma + max), forward with folded BatchNorm statistics and weight gradient, patch and consecutive-tile
    forms, ragged maps (pixels outside the map are zero padding of the ACTIVATION, not relu(shift))."""
    y = rnd(N, H, W, C, seed=1).to(DEV)
    sc, sh = (rnd(C, seed=2) * 0.5 + 1).to(DEV), rnd(C, seed=3).to(DEV)
    z = torch.full_like(y, float('nan'))
    _lib.check(L().dbn_bn_apply_t(0, y.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, None, None, z.data_ptr(), N * H * W, C, 1, stream()), 'bn apply')
    w = rnd(Co, C, 3, 3, seed=4, scale=(2.0 / (C * 9))**0.5).to(DEV)
    up = torch.empty(L().dbn_winograd_panel_floats(Co, C), device=DEV)
    _lib.check(L().dbn_winograd_pack(w.data_ptr(), Co, C, C, 0, up.data_ptr(), stream()), 'pack')
    assert L().dbn_winograd_eligible(N, H, W, C, Co)
    res = []
    for src, a, b in ((z, None, None), (y, sc.data_ptr(), sh.data_ptr())):
        out = torch.full((N, H, W, Co), float('nan'), device=DEV)
        gamma, beta = torch.ones(Co, device=DEV), torch.zeros(Co, device=DEV)
        rm, rv = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
        o4 = [torch.full((Co, ), float('nan'), device=DEV) for _ in range(4)]
        ws = torch.empty(L().dbn_winograd_ws_floats(N, H, W, Co), device=DEV)
        _lib.check(L().dbn_winograd_conv_bn_act_f32(src.data_ptr(), a, b, up.data_ptr(), None, out.data_ptr(), N, H, W, C, Co, gamma.data_ptr(),
                                                    beta.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), *[t.data_ptr() for t in o4],
                                                    ws.data_ptr(), stream()), 'winograd act')
        res.append([out] + o4 + [rm, rv])
    if L().dbn_winograd_wgrad_eligible(N, H, W, Co, C, C):
        dy = rnd(N, H, W, Co, seed=5).to(DEV)
        for i, (src, a, b) in enumerate(((z, None, None), (y, sc.data_ptr(), sh.data_ptr()))):
            slab = torch.empty(L().dbn_winograd_wgrad_slab_floats(N, H, W, Co, C), device=DEV)
            g = torch.full((Co, C, 3, 3), float('nan'), device=DEV)
            _lib.check(L().dbn_winograd_wgrad_f32(3, dy.data_ptr(), src.data_ptr(), a, b, slab.data_ptr(), g.data_ptr(), N, H, W, Co, C, C, 1.0,
                                                  stream()), 'winograd wgrad act')
            res[i].append(g)
    torch.cuda.synchronize()
    assert len(res[0]) == len(res[1])
    for t0, t1 in zip(*res):
        assert bool(torch.isfinite(t0).all()) and torch.equal(t0, t1)
    # and the activation really matters (relu(shift) != 0 at the padding would show here)
    assert float((z - y).abs().max()) > 0.1


@pytest.mark.parametrize('at', [0, 1, 2])
@pytest.mark.parametrize('case', [(2, 64, 64, 3, 1, 1, 16, 32, True), (1, 64, 128, 3, 2, 1, 18, 14, False), (2, 64, 128, 1, 2, 0, 16, 16, False),
                                  (1, 256, 64, 1, 1, 0, 12, 12, True), (2, 128, 128, 3, 1, 1, 8, 16, True)])
def test_inference_epilogue_folded_bn_residual_relu(case, at):
    """Round 5: dbn_fold_bn_eval + dbn_igemm_act_t / dbn_winograd_conv_act_f32 — relu(bn_eval(conv(x) + bias) [+ residual]) in ONE launch
    (basic.py:32-36, resnet.py:70-91 under model.eval()) against F.conv2d -> F.batch_norm(training=False) -> add -> relu in fp64 on the
    operands as stored.  fp32 (direct and Winograd kernel), bf16 and fp16 storage (output rounding 2^-9 / 2^-11 relative)."""
    N, Ci, Co, k, s, p, H, W, with_res = case
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, k, k, seed=2, scale=(2.0 / (Ci * k * k))**0.5)
    b = rnd(Co, seed=3) * 0.1
    gamma, beta = rnd(Co, seed=4) * 0.3 + 1, rnd(Co, seed=5) * 0.2
    rm, rv = rnd(Co, seed=6) * 0.1, torch.rand(Co, generator=torch.Generator().manual_seed(7)) + 0.5
    dt = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}[at]
    xs = nhwc(x).to(dt)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = rnd(N, Co, Ho, Wo, seed=8) if with_res else None
    rs_ = nhwc(res).to(dt) if with_res else None
    wd, bd = w.to(DEV), b.to(DEV)
    gd, btd, rmd, rvd = gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV)  # (kept alive: the call takes raw pointers)
    wf, bf = torch.full_like(wd, float('nan')), torch.full((Co, ), float('nan'), device=DEV)
    _lib.check(L().dbn_fold_bn_eval(wd.data_ptr(), Co, Ci * k * k, bd.data_ptr(), gd.data_ptr(), btd.data_ptr(), rmd.data_ptr(), rvd.data_ptr(),
                                    1e-5, wf.data_ptr(), bf.data_ptr(), stream()), 'fold')
    sc = gamma.double() / torch.sqrt(rv.double() + 1e-5)
    report('folded weight', wf.cpu(), w.double() * sc.view(-1, 1, 1, 1), 1e-6, 1e-6)
    report('folded bias', bf.cpu(), beta.double() + (b.double() - rm.double()) * sc, 1e-6, 1e-6)
    # reference on the operands as stored
    xr = nchw(xs.float()).double()
    ref = F.batch_norm(F.conv2d(xr, w.double(), b.double(), s, p), rm.double(), rv.double(), gamma.double(), beta.double(), False, 0.1, 1e-5)
    if with_res:
        ref = ref + nchw(rs_.float()).double()
    ref = torch.relu(ref)
    ns, kind = (0, 0) if at == 0 else (1, at)
    n = L().dbn_igemm_panel_floats_t(kind, Co, Ci, k, k, 0, 1, Ci)
    wpk = torch.empty(n, device=DEV)
    _lib.check(L().dbn_pack_weights_t(kind, wf.data_ptr(), Co, Ci, k, k, 0, 1, Ci, wpk.data_ptr(), stream()), 'pack')
    y = torch.full((N, Ho, Wo, Co), float('nan'), device=DEV, dtype=dt)
    _lib.check(L().dbn_igemm_act_t(at, ns, xs.data_ptr(), wpk.data_ptr(), bf.data_ptr(), rs_.data_ptr() if with_res else None, 1, y.data_ptr(),
                                   N, H, W, Ci, Ho, Wo, Co, k, k, s, p, 0, 0, stream()), 'igemm act')
    scale = float(ref.abs().max())
    # (16-bit: the folded weights are rounded to the storage type once more than the unfolded chain's: 2^-8 / 2^-10 of the output scale)
    tol = {0: 1e-5, 1: 2.0**-7, 2: 2.0**-9}[at]
    report('igemm act %s at %d' % (case, at), nchw(y.float()), ref, tol * scale, tol)
    assert float(y.float().min()) >= 0.0
    if at == 0 and k == 3 and s == 1 and L().dbn_winograd_eligible(N, H, W, Ci, Co):
        up = torch.empty(L().dbn_winograd_panel_floats(Co, Ci), device=DEV)
        _lib.check(L().dbn_winograd_pack(wf.data_ptr(), Co, Ci, Ci, 0, up.data_ptr(), stream()), 'winograd pack')
        for persistent in (0, 1):
            L().dbn_set_winograd_persistent(persistent)
            yw = torch.full_like(y, float('nan'))
            _lib.check(L().dbn_winograd_conv_act_f32(xs.data_ptr(), up.data_ptr(), bf.data_ptr(), rs_.data_ptr() if with_res else None, 1,
                                                     yw.data_ptr(), N, H, W, Ci, Co, stream()), 'winograd act')
            report('winograd act %s' % (case, ), nchw(yw), ref, 1e-5 * scale, 1e-5)
        L().dbn_set_winograd_persistent(0)


@pytest.mark.parametrize('N,Ci,Co,H,W', [(16, 64, 64, 96, 96), (8, 128, 128, 80, 80), (40, 256, 256, 40, 40)])
def test_winograd_persistent_forms_are_bit_identical(N, Ci, Co, H, W):
    """Round 5: the persistent forms of winograd_f32_kernel (workgroups that pull (patch, channel tile) items from per-XCD counters;
    a static schedule) compute every item exactly as the one-workgroup-per-item form does: same bits, with the train-mode BatchNorm
    statistics, on launches of more than 512 items (fewer: the launch is not persistent).  Repeated: the counters are left at zero."""
    x = torch.randn(N, H, W, Ci, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    w = rnd(Co, Ci, 3, 3, seed=2, scale=(2.0 / (Ci * 9))**0.5).to(DEV)
    up = torch.empty(L().dbn_winograd_panel_floats(Co, Ci), device=DEV)
    _lib.check(L().dbn_winograd_pack(w.data_ptr(), Co, Ci, Ci, 0, up.data_ptr(), stream()), 'winograd pack')
    assert L().dbn_winograd_rows(N, H, W) * (Co // 64) > 512
    outs = []
    for mode in (0, 1, 2, 1):
        L().dbn_set_winograd_persistent(mode)
        gamma, beta = torch.ones(Co, device=DEV), torch.zeros(Co, device=DEV)
        rm, rv = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
        sc, sh, mu, rs = (torch.full((Co, ), float('nan'), device=DEV) for _ in range(4))
        ws = torch.full((L().dbn_winograd_ws_floats(N, H, W, Co), ), float('nan'), device=DEV)
        y = torch.full((N, H, W, Co), float('nan'), device=DEV)
        _lib.check(L().dbn_winograd_conv_bn_f32(x.data_ptr(), up.data_ptr(), None, y.data_ptr(), N, H, W, Ci, Co, gamma.data_ptr(), beta.data_ptr(),
                                                1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(),
                                                rs.data_ptr(), ws.data_ptr(), stream()), 'winograd+bn')
        torch.cuda.synchronize()
        outs.append((y, sc, sh, mu, rs))
    L().dbn_set_winograd_persistent(0)
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


@pytest.mark.parametrize('at', [1, 2])
@pytest.mark.parametrize('N,H,W', [(2, 64, 64), (1, 100, 70), (3, 36, 47), (2, 132, 250), (1, 28, 600)])
def test_stem_conv_bn_relu_maxpool_in_one_launch_on_16bit_storage(N, H, W, at):
    """Round 5, csrc/stem16.hip stem7x7_pool_b16_kernel (inference): MaxPool2d(3, 2, 1)(relu(bn_eval(conv7x7/2(x)))) of resnet.py:231-235 in
    ONE launch on the packed zero-bordered image — against F.max_pool2d(F.relu(F.conv2d(...) * scale + shift)) in fp64 on the operands as
    stored, and against the two-kernel path (dbn_stem16_conv_bn_t + dbn_bnrelu_maxpool_fwd_t), which rounds the conv output to the storage
    type before the BatchNorm.  Strips of 15 pooled columns (ragged last strip, several strips), several row chunks, odd widths."""
    dt = {1: torch.bfloat16, 2: torch.float16}[at]
    x = rnd(N, 3, H, W, seed=1)
    w = rnd(64, 3, 7, 7, seed=2, scale=(2.0 / 147)**0.5)
    sc, sh = rnd(64, seed=3) * 0.4 + 1.0, rnd(64, seed=4) * 0.5
    xd, wd, scd, shd = x.to(DEV), w.to(DEV), sc.to(DEV), sh.to(DEV)
    Hp, Wp = L().dbn_stem16_padded_h(H), L().dbn_stem16_padded_w(W)
    xp = torch.zeros(N, Hp, Wp, 4, device=DEV, dtype=dt)
    _lib.check(L().dbn_nchw3_to_padded4_t(at, xd.data_ptr(), xp.data_ptr(), None, N, H, W, stream()), 'padded4')
    panel = torch.empty(L().dbn_stem16_panel_bytes(), device=DEV, dtype=torch.uint8)
    _lib.check(L().dbn_stem16_pack(at, wd.data_ptr(), panel.data_ptr(), stream()), 'stem16 pack')
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    assert L().dbn_stem16_pool_eligible(at, N, H, W) == 1 and L().dbn_stem16_pool_eligible(at, N, H + 2, W) == 0  # (odd conv height: two launches)
    Hq, Wq = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
    conv = F.conv2d(x.to(dt).double(), w.to(dt).double(), None, 2, 3)
    ref = F.max_pool2d(F.relu(conv * sc.double().view(1, 64, 1, 1) + sh.double().view(1, 64, 1, 1)), 3, 2, 1)
    assert ref.shape == (N, 64, Hq, Wq)
    out = torch.full((N, Hq, Wq, 64), float('nan'), device=DEV, dtype=dt)
    _lib.check(L().dbn_stem16_conv_bn_relu_pool_t(at, xp.data_ptr(), panel.data_ptr(), scd.data_ptr(), shd.data_ptr(), out.data_ptr(), N, H, W,
                                                  stream()), 'stem16 pool')
    tol = {1: 2.0**-8, 2: 2.0**-10}[at]
    scale = float(ref.abs().max())
    report('stem conv+bn+relu+pool at %d' % at, nchw(out.float()), ref, tol * scale * 0.5 + 1e-6, tol)
    # the two-kernel path: the same numbers up to the extra rounding of the conv output to the storage type
    y = torch.empty((N, Ho, Wo, 64), device=DEV, dtype=dt)
    _lib.check(L().dbn_stem16_conv_bn_t(at, xp.data_ptr(), panel.data_ptr(), y.data_ptr(), N, H, W, None, None, 0.0, 0.0, None, None, None, None,
                                        None, None, None, stream()), 'stem16 conv')
    out2 = torch.full_like(out, float('nan'))
    _lib.check(L().dbn_bnrelu_maxpool_fwd_t(at, y.data_ptr(), scd.data_ptr(), shd.data_ptr(), out2.data_ptr(), N, Ho, Wo, 64, stream()), 'maxpool')
    report('one launch vs conv + pool', out.float().cpu(), out2.float().cpu(), 3 * tol * scale, 2 * tol)
    # bit-identical from run to run
    out3 = torch.full_like(out, float('nan'))
    _lib.check(L().dbn_stem16_conv_bn_relu_pool_t(at, xp.data_ptr(), panel.data_ptr(), scd.data_ptr(), shd.data_ptr(), out3.data_ptr(), N, H, W,
                                                  stream()), 'stem16 pool')
    assert torch.equal(out, out3)


@pytest.mark.parametrize('at', [1, 2])
@pytest.mark.parametrize('N,H,W', [(2, 64, 64), (1, 96, 70), (3, 33, 47), (2, 128, 160)])
def test_stem_conv_on_packed_16bit_input(N, H, W, at):
    """Round 5, csrc/stem16.hip: the stem Conv2d(3 -> 64, 7x7, stride 2, pad 3) of resnet.py:167-172,231-235 in 16-bit storage on the packed,
    zero-bordered 4-channel image (dbn_nchw3_to_padded4_t + dbn_stem16_pack + dbn_stem16_conv_bn_t) against F.conv2d in fp64 on the
    operands as stored (image and weights rounded to the storage type): what is left is fp32 accumulation and the output rounding
    (2^-9 bf16 / 2^-11 fp16 relative); the train-mode BatchNorm statistics — taken from the fp32 accumulators — against F.batch_norm.
    Odd sizes: Ho = (H - 1) / 2 + 1, partial last 32-row block."""
    dt = {1: torch.bfloat16, 2: torch.float16}[at]
    x = rnd(N, 3, H, W, seed=1)
    w = rnd(64, 3, 7, 7, seed=2, scale=(2.0 / 147)**0.5)
    xd, wd = x.to(DEV), w.to(DEV)
    Hp, Wp = L().dbn_stem16_padded_h(H), L().dbn_stem16_padded_w(W)
    xp = torch.zeros(N, Hp, Wp, 4, device=DEV, dtype=dt)
    x4 = torch.full((N, H, W, 4), float('nan'), device=DEV, dtype=dt)
    _lib.check(L().dbn_nchw3_to_padded4_t(at, xd.data_ptr(), xp.data_ptr(), x4.data_ptr(), N, H, W, stream()), 'padded4')
    xs = x.to(dt).float()  # the image as stored
    assert torch.equal(x4[..., :3].float().cpu(), xs.permute(0, 2, 3, 1)) and float(x4[..., 3].abs().max()) == 0
    assert torch.equal(xp[:, 3:H + 3, 3:W + 3, :3].float().cpu(), xs.permute(0, 2, 3, 1))
    assert float(xp[:, :3].abs().max()) == 0 and float(xp[:, H + 3:].abs().max()) == 0 and float(xp[:, :, :3].abs().max()) == 0 and float(xp[:, :, W + 3:].abs().max()) == 0
    panel = torch.empty(L().dbn_stem16_panel_bytes(), device=DEV, dtype=torch.uint8)
    _lib.check(L().dbn_stem16_pack(at, wd.data_ptr(), panel.data_ptr(), stream()), 'stem16 pack')
    assert L().dbn_stem16_eligible(at, N, H, W)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    ref = F.conv2d(xs.double(), w.to(dt).double(), None, 2, 3)
    assert ref.shape == (N, 64, Ho, Wo)
    scale = float(ref.abs().max())
    tol = {1: 2.0**-8, 2: 2.0**-10}[at]
    y = torch.full((N, Ho, Wo, 64), float('nan'), device=DEV, dtype=dt)
    _lib.check(L().dbn_stem16_conv_bn_t(at, xp.data_ptr(), panel.data_ptr(), y.data_ptr(), N, H, W, None, None, 0.0, 0.0, None, None, None, None,
                                        None, None, None, stream()), 'stem16 conv')
    report('stem16 conv at %d' % at, nchw(y.float()), ref, tol * scale, tol)
    # with the train-mode BatchNorm statistics
    gamma, beta = (rnd(64, seed=4) * 0.3 + 1).to(DEV), rnd(64, seed=5).to(DEV)
    rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    sc, sh, mu, rs = (torch.full((64, ), float('nan'), device=DEV) for _ in range(4))
    ws = torch.full(((3 * 64 + 1) * L().dbn_stem16_rows(), ), float('nan'), device=DEV)
    y2 = torch.full_like(y, float('nan'))
    _lib.check(L().dbn_stem16_conv_bn_t(at, xp.data_ptr(), panel.data_ptr(), y2.data_ptr(), N, H, W, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1,
                                        rm.data_ptr(), rv.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), ws.data_ptr(),
                                        stream()), 'stem16 conv+bn')
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    mean, var = ref.mean((0, 2, 3)), ref.var((0, 2, 3), unbiased=False)
    report('stem16 BN mean', mu.cpu(), mean, 1e-5 * scale, 1e-5)
    report('stem16 BN rstd', rs.cpu(), 1.0 / torch.sqrt(var + 1e-5), 1e-5, 1e-4)
    report('stem16 BN scale', sc.cpu(), gamma.cpu().double() / torch.sqrt(var + 1e-5), 1e-5, 1e-4)
    n = N * Ho * Wo
    report('stem16 BN running var', rv.cpu(), 0.9 + 0.1 * var * n / (n - 1), 1e-6, 1e-4)


@pytest.mark.parametrize('at', [1, 2])
@pytest.mark.parametrize('N,H,W,with_bias', [(2, 16, 16, True), (1, 24, 17, False), (3, 5, 7, True), (2, 40, 40, True)])
def test_conv_transpose_2x2_16bit_kernel(N, H, W, with_bias, at):
    """Round 5, csrc/convt16.hip: ConvTranspose2d(64 -> 64, 2x2, stride 2) forward in 16-bit storage (segmentation_head.py:27-29,74-76) with
    its own kernel — weight panel in registers, A fragments straight from global memory, LDS-transposed 16-byte stores — against
    F.conv_transpose2d in fp64 on the operands as stored, and its train-mode BatchNorm statistics against F.batch_norm; equal to the
    generic parity-class launch up to the output rounding.  Odd sizes: partial last 32-pixel block."""
    dt = {1: torch.bfloat16, 2: torch.float16}[at]
    x = rnd(N, 64, H, W, seed=1)
    w = rnd(64, 64, 2, 2, seed=2, scale=(2.0 / 64)**0.5)
    b = rnd(64, seed=3) * 0.1 if with_bias else None
    xs = nhwc(x).to(dt)
    wd = w.to(DEV)
    bd = b.to(DEV) if with_bias else None
    panel = torch.empty(L().dbn_convt16_panel_bytes(), device=DEV, dtype=torch.uint8)
    _lib.check(L().dbn_convt16_pack(at, wd.data_ptr(), panel.data_ptr(), stream()), 'convt16 pack')
    assert L().dbn_convt16_eligible(at, N, H, W, 64, 64)
    ref = F.conv_transpose2d(nchw(xs.float()).double(), w.to(dt).double(), b.double() if with_bias else None, 2)
    scale = float(ref.abs().max())
    tol = {1: 2.0**-8, 2: 2.0**-10}[at]
    y = torch.full((N, 2 * H, 2 * W, 64), float('nan'), device=DEV, dtype=dt)
    _lib.check(L().dbn_convt16_bn_t(at, xs.data_ptr(), panel.data_ptr(), bd.data_ptr() if with_bias else None, y.data_ptr(), N, H, W, None, None,
                                    0.0, 0.0, None, None, None, None, None, None, None, stream()), 'convt16')
    report('convt16 at %d' % at, nchw(y.float()), ref, tol * scale, tol)
    gamma, beta = (rnd(64, seed=4) * 0.3 + 1).to(DEV), rnd(64, seed=5).to(DEV)
    rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    sc, sh, mu, rs = (torch.full((64, ), float('nan'), device=DEV) for _ in range(4))
    ws = torch.full(((3 * 64 + 1) * L().dbn_convt16_rows(), ), float('nan'), device=DEV)
    y2 = torch.full_like(y, float('nan'))
    _lib.check(L().dbn_convt16_bn_t(at, xs.data_ptr(), panel.data_ptr(), bd.data_ptr() if with_bias else None, y2.data_ptr(), N, H, W,
                                    gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                    mu.data_ptr(), rs.data_ptr(), ws.data_ptr(), stream()), 'convt16+bn')
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    mean, var = ref.mean((0, 2, 3)), ref.var((0, 2, 3), unbiased=False)
    report('convt16 BN mean', mu.cpu(), mean, 1e-5 * scale, 1e-5)
    report('convt16 BN rstd', rs.cpu(), 1.0 / torch.sqrt(var + 1e-5), 1e-5, 1e-4)
    n = N * 4 * H * W
    report('convt16 BN running var', rv.cpu(), 0.9 + 0.1 * var * n / (n - 1), 1e-6, 1e-4)


@pytest.mark.parametrize('at', [0, 1])
@pytest.mark.parametrize('N,H,W,ties', [(2, 16, 16, False), (1, 17, 23, False), (3, 32, 48, True), (2, 9, 14, True)])
def test_stem_pool_with_recorded_argmax_and_its_backward_through_the_batchnorm(N, H, W, ties, at):
    """Late in round 5, csrc/pointwise.hip (dbn_bnrelu_maxpool_fwd_arg_t, dbn_maxpool_bn_backward_t): MaxPool2d(3, 2, 1) over
    relu(bn1(y)) of resnet.py:231-235 with the window's FIRST maximum recorded, and the gradient at the conv output y through pool, ReLU and the
    train-mode BatchNorm in one pass — against torch's autograd in fp64 (nn.MaxPool2d routes a window's gradient to its first maximum).
    ties: y quantised to a few levels, so that most windows hold several equal maxima (and whole windows of zeros after the ReLU) —
    the case in which round 4's kernel pair (a gradient to EVERY tying position) differs from the reference."""
    C = 64
    dt = {0: torch.float32, 1: torch.bfloat16}[at]
    y = rnd(N, C, H, W, seed=1)
    if ties:
        y = torch.round(y * 2) / 2
    y = y.to(dt).double()  # the operand as stored
    gamma, beta = (rnd(C, seed=2) * 0.3 + 1).double(), (rnd(C, seed=3) * 0.3).double()
    if ties:
        gamma[::7] = -gamma[::7]  # negative scales: the maximum of z is not the maximum of y
    yr = y.clone().requires_grad_(True)
    mean = yr.mean((0, 2, 3), keepdim=True)
    var = yr.var((0, 2, 3), unbiased=False, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    z = F.relu((yr - mean) * rstd * gamma.view(1, C, 1, 1) + beta.view(1, C, 1, 1))
    pool_ref = F.max_pool2d(z, 3, 2, 1)
    dp = rnd(*pool_ref.shape, seed=4).to(dt).double()
    dy_ref, = torch.autograd.grad(pool_ref, yr, dp)
    xh = ((y - mean) * rstd).detach()
    # forward operands exactly as the engine hands them over: scale / shift (fp32), saved mean / rstd (fp32)
    meanf, rstdf = mean.detach().view(C).float(), rstd.detach().view(C).float()
    scale = (gamma.float() * rstdf)
    shift = beta.float() - meanf * scale
    Ho, Wo = pool_ref.shape[2:]
    ys = nhwc(y.float()).to(dt)
    scd, shd, meand, rstdd, gammad = scale.to(DEV), shift.to(DEV), meanf.to(DEV), rstdf.to(DEV), gamma.float().to(DEV)
    pool = torch.full((N, Ho, Wo, C), float('nan'), device=DEV, dtype=dt)
    ypool = torch.full((N, Ho, Wo, C), float('nan'), device=DEV, dtype=dt)
    idx = torch.full((N, Ho, Wo, C), 77, device=DEV, dtype=torch.uint8)
    _lib.check(L().dbn_bnrelu_maxpool_fwd_arg_t(at, ys.data_ptr(), scd.data_ptr(), shd.data_ptr(), pool.data_ptr(), idx.data_ptr(),
                                                ypool.data_ptr(), N, H, W, C, stream()), 'pool fwd arg')
    tol = {0: 1e-5, 1: 2.0**-7}[at]
    report('pool fwd (argmax form)', nchw(pool.float()), pool_ref.detach(), tol, tol)
    # the same pooled values as the kernel without the record
    pool2 = torch.full_like(pool, float('nan'))
    _lib.check(L().dbn_bnrelu_maxpool_fwd_t(at, ys.data_ptr(), scd.data_ptr(), shd.data_ptr(), pool2.data_ptr(), N, H, W, C, stream()),
               'pool fwd')
    assert torch.equal(pool, pool2)
    codes = idx.cpu()
    assert int(((codes > 8) & (codes != 15)).sum()) == 0
    assert bool(((codes == 15) == (pool.float().cpu() == 0)).all())
    # the recorded position holds the recorded y, and that y gives the pooled value
    yc = y.float()
    n_, oh, ow, c_ = torch.meshgrid(torch.arange(N), torch.arange(Ho), torch.arange(Wo), torch.arange(C), indexing='ij')
    live = codes != 15
    ih = (2 * oh - 1 + (codes // 3).long())[live]
    iw = (2 * ow - 1 + (codes % 3).long())[live]
    assert bool(((ih >= 0) & (ih < H) & (iw >= 0) & (iw < W)).all())
    assert torch.equal(yc[n_[live], c_[live], ih, iw], ypool.float().cpu()[live])
    # backward
    dps = nhwc(dp.float()).to(dt)
    dy = torch.full((N, H, W, C), float('nan'), device=DEV, dtype=dt)
    dgamma, dbeta = torch.full((C, ), float('nan'), device=DEV), torch.full((C, ), float('nan'), device=DEV)
    ws = torch.full((L().dbn_maxpool_bn_backward_ws_floats(N, H, W, C), ), float('nan'), device=DEV)
    _lib.check(L().dbn_maxpool_bn_backward_t(at, ys.data_ptr(), dps.data_ptr(), idx.data_ptr(), ypool.data_ptr(), meand.data_ptr(),
                                             rstdd.data_ptr(), gammad.data_ptr(), dy.data_ptr(), dgamma.data_ptr(),
                                             dbeta.data_ptr(), N, H, W, C, 1.0, ws.data_ptr(), stream()), 'pool + bn bwd')
    g_ref, = torch.autograd.grad(F.max_pool2d(z, 3, 2, 1), z, dp, retain_graph=True)
    g_ref = (g_ref * (z > 0)).detach()
    scale_g = float(dy_ref.abs().max())
    btol = {0: 2e-5, 1: 2.0**-7}[at]
    report('dy through pool, relu, bn', nchw(dy.float()), dy_ref, btol * scale_g, btol)
    report('dgamma', dgamma.cpu(), (g_ref * xh).sum((0, 2, 3)), 1e-4 * float((g_ref * xh).sum((0, 2, 3)).abs().max()) + 1e-5, 1e-4)
    report('dbeta', dbeta.cpu(), g_ref.sum((0, 2, 3)), 1e-4 * float(g_ref.sum((0, 2, 3)).abs().max()) + 1e-5, 1e-4)
    # run-to-run bit identity
    dy2 = torch.full_like(dy, float('nan'))
    _lib.check(L().dbn_maxpool_bn_backward_t(at, ys.data_ptr(), dps.data_ptr(), idx.data_ptr(), ypool.data_ptr(), meand.data_ptr(),
                                             rstdd.data_ptr(), gammad.data_ptr(), dy2.data_ptr(), dgamma.data_ptr(),
                                             dbeta.data_ptr(), N, H, W, C, 1.0, ws.data_ptr(), stream()), 'pool + bn bwd')
    assert torch.equal(dy, dy2)
