"""-m gpu: the weight-resident 3x3 / stride-1 kernel of the 16-bit storage types (csrc/wres16.hip, conv3x3_wres16_kernel), through the
C ABI it is reached by (dbn_igemm_t, dbn_igemm_act_t, dbn_conv_bn_t, dbn_igemm_bnsums_t): the convs of
/root/reference/src/modules/resnet.py:70-91 (layer1 / layer2), segmentation_body.py:55-61 and segmentation_head.py:24-29,64-68 with
64 -> 64, 128 -> 128 and 256 -> 64 channels, forward and data gradient.

Yardsticks: F.conv2d in fp64 on the operands AS STORED (the kernel sums in fp32 and rounds once), and the pixel-patch kernel it replaces
(dbn_set_wres16(0)) — another summation order of the same products, so the two agree to a rounding step of the storage type, not bit for
bit.  Run-to-run the kernel must be bit-reproducible (fixed summation order, no atomics on data)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from gpu_util import DEV, L, nchw, nhwc, report, rnd, stream
from db_text_minimal_amd import _lib
from test_ops_gpu import AT_OF, igemm_t, pack_t

pytestmark = pytest.mark.gpu

ULP = {torch.bfloat16: 2.0**-7, torch.float16: 2.0**-10}  # spacing of the storage type relative to the binade start


@pytest.fixture
def any_width(request):
    """Maps whose width is not a multiple of 32 go to the pixel-patch kernel by default (faster there); dbn_set_wres16(2) sends them to
    the weight-resident kernel all the same, so that its ragged last strip stays tested."""
    L().dbn_set_wres16(2)
    yield
    L().dbn_set_wres16(1)


def _is_wres(dtype, mode, N, H, W, Cs, Cd):
    return bool(L().dbn_igemm_kernel_config(AT_OF[dtype], 1, mode, N, H, W, Cs, H, W, Cd, 3, 3, 1, 1, 0, 1) & 64)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('Ci,Co', [(64, 64), (128, 128), (256, 64)])
@pytest.mark.parametrize('N,H,W', [(2, 24, 64), (3, 16, 80), (1, 40, 160), (5, 8, 32)])
def test_forward_and_data_gradient_vs_fp64_and_the_patch_kernel(dtype, Ci, Co, N, H, W, any_width):
    """Forward (bias), data gradient (accumulate onto a base tensor) and the inference epilogue (bias + residual + ReLU): every strip
    layout (W = 32 k, a ragged last strip at W = 80), workgroup ranges that cross strips and images, the K-split exchange (Ci > 64)."""
    kind = 2 if dtype == torch.float16 else 1
    x = nhwc(rnd(N, Ci, H, W, seed=11)).to(dtype)
    w = rnd(Co, Ci, 3, 3, seed=12, scale=(2.0 / (Ci * 9))**0.5)
    b = rnd(Co, seed=13).to(DEV)
    res = nhwc(rnd(N, Co, H, W, seed=16)).to(dtype)
    # data gradient of the conv with the channel pair REVERSED as the kernel sees it: dy has Cs channels, dx has Cd
    Cs1, Cd1 = Ci, Co  # (mode 1 on the same (Cs, Cd) pair: a conv Co_fwd = Cs1 -> ... i.e. weights [Cs1][Cd1])
    w1 = rnd(Cs1, Cd1, 3, 3, seed=17, scale=(2.0 / (Cs1 * 9))**0.5)
    dy = nhwc(rnd(N, Cs1, H, W, seed=14)).to(dtype)
    base = nhwc(rnd(N, Cd1, H, W, seed=15)).to(dtype)
    assert _is_wres(dtype, 0, N, H, W, Ci, Co) and _is_wres(dtype, 1, N, H, W, Cs1, Cd1), 'the weight-resident kernel was not selected'

    def run():
        y = torch.full((N, H, W, Co), float('nan'), device=DEV, dtype=dtype)
        igemm_t(x, pack_t(w, 0, 1, kind, Ci), b, y, 3, 1, 1, 0)
        d = base.clone()
        igemm_t(dy, pack_t(w1, 1, 1, kind), None, d, 3, 1, 1, 1, accumulate=1)
        a = torch.full((N, H, W, Co), float('nan'), device=DEV, dtype=dtype)
        _lib.check(L().dbn_igemm_act_t(AT_OF[dtype], 1, x.data_ptr(), pack_t(w, 0, 1, kind, Ci).data_ptr(), b.data_ptr(), res.data_ptr(), 1,
                                       a.data_ptr(), N, H, W, Ci, H, W, Co, 3, 3, 1, 1, 0, 0, stream()), 'igemm_act_t')
        torch.cuda.synchronize()
        return y, d, a

    y1, d1, a1 = run()
    y2, d2, a2 = run()
    assert torch.equal(y1, y2) and torch.equal(d1, d2) and torch.equal(a1, a2), 'not bit-reproducible run to run'
    try:
        assert L().dbn_set_wres16(0) == 2
        assert not _is_wres(dtype, 0, N, H, W, Ci, Co)
        y0, d0, a0 = run()
    finally:
        L().dbn_set_wres16(2)
    wq = w.to(dtype).double()
    ref_y = F.conv2d(nchw(x.double()), wq, b.double().cpu(), 1, 1)
    xg = torch.zeros(N, Cd1, H, W, dtype=torch.float64, requires_grad=True)
    (ref_d, ) = torch.autograd.grad(F.conv2d(xg, w1.to(dtype).double(), None, 1, 1), xg, nchw(dy.double()))
    ref_d = ref_d + nchw(base.double())
    ref_a = torch.relu(ref_y + nchw(res.double()))
    u = ULP[dtype]
    for tag, got, old, ref in (('forward', y1, y0, ref_y), ('data gradient', d1, d0, ref_d), ('act', a1, a0, ref_a)):
        assert torch.isfinite(got.float()).all(), tag
        g, o = nchw(got.double()), nchw(old.double())
        # one rounding of the storage type (half an ulp of the value) + fp32 summation noise
        report('%s vs fp64' % tag, g, ref, 2e-5 * float(ref.abs().max()), 0.51 * u)
        report('%s vs pixel-patch kernel' % tag, g, o, 4e-5 * float(ref.abs().max()), 1.01 * u)


@pytest.mark.parametrize('Ci,Co', [(64, 64), (128, 128), (256, 64)])
@pytest.mark.parametrize('N,H,W,accumulate', [(2, 24, 64, 0), (3, 16, 80, 0), (4, 8, 96, 1)])
def test_train_mode_batchnorm_statistics_epilogue(Ci, Co, N, H, W, accumulate, any_width):
    """dbn_conv_bn_t on bf16 storage: the conv's epilogue accumulates the train-mode BatchNorm statistics of its output — one partial
    row per workgroup (pivot = its first pixel), the rows a pixel-patch launch would have written beyond that left empty — and the
    finalize kernel merges them: scale / shift / saved mean / rstd / running statistics against fp64 on the fp32 accumulators' values
    (= the fp64 conv to fp32 summation noise) and against the pixel-patch kernel."""
    bf = torch.bfloat16
    x = nhwc(rnd(N, Ci, H, W, seed=1) * 1.5 + 0.3).to(bf)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=(2.0 / (Ci * 9))**0.5)
    bias = rnd(Co, seed=4).to(DEV)
    old = nhwc(rnd(N, Co, H, W, seed=9)).to(bf)
    gam, bet = (rnd(Co, seed=5) * 0.2 + 1).to(DEV), rnd(Co, seed=6).to(DEV)

    def run():
        y = old.clone() if accumulate else torch.full((N, H, W, Co), float('nan'), device=DEV, dtype=bf)
        rm_, rv_ = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
        sc, sh, mu, rs = (torch.full((Co, ), float('nan'), device=DEV) for _ in range(4))
        ws = torch.full((L().dbn_conv_bn_ws_floats(N, H, W, Co, 0, 1), ), float('nan'), device=DEV)
        _lib.check(L().dbn_conv_bn_t(1, x.data_ptr(), pack_t(w, 0, 1, 1, Ci).data_ptr(), bias.data_ptr(), y.data_ptr(), N, H, W, Ci, H, W, Co,
                                     3, 3, 1, 1, 0, accumulate, 0, 1, gam.data_ptr(), bet.data_ptr(), 1e-5, 0.1, rm_.data_ptr(), rv_.data_ptr(),
                                     sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), ws.data_ptr(), stream()), 'conv_bn_t')
        torch.cuda.synchronize()
        return dict(y=y, scale=sc, shift=sh, mean=mu, rstd=rs, run_mean=rm_, run_var=rv_)

    assert _is_wres(bf, 0, N, H, W, Ci, Co)
    a, a2 = run(), run()
    for k in a:
        assert torch.equal(a[k], a2[k]), 'not bit-reproducible: ' + k
    try:
        L().dbn_set_wres16(0)
        o = run()
    finally:
        L().dbn_set_wres16(2)
    ref = F.conv2d(nchw(x.double()), w.to(bf).double(), bias.double().cpu(), 1, 1) + (nchw(old.double()) if accumulate else 0)
    mean, var = ref.mean((0, 2, 3)), ref.var((0, 2, 3), unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    report('saved mean', a['mean'].cpu(), mean, 1e-5, 1e-5)
    report('saved rstd', a['rstd'].cpu(), rstd, 1e-6, 2e-5)
    report('scale', a['scale'].cpu(), gam.double().cpu() * rstd, 1e-6, 2e-5)
    report('shift', a['shift'].cpu(), bet.double().cpu() - mean * gam.double().cpu() * rstd, 2e-5, 2e-5)
    n = N * H * W
    report('running var', a['run_var'].cpu(), 0.9 + 0.1 * var * n / (n - 1), 1e-6, 2e-5)
    for k in ('scale', 'shift', 'mean', 'rstd', 'run_mean', 'run_var'):
        report('%s vs the pixel-patch kernel' % k, a[k].cpu(), o[k].cpu(), 2e-6, 2e-6)
    report('y vs the pixel-patch kernel', a['y'].float().cpu(), o['y'].float().cpu(), 4e-5 * float(ref.abs().max()), 1.01 * ULP[bf])


@pytest.mark.parametrize('geom', [(128, 128), (256, 64)])
def test_full_size_launches_are_bit_reproducible_and_cover_every_pixel(geom):
    """BASELINE's sizes (16 x 160 x 160 / 16 x 80 x 80, bf16): 1024 / 256 workgroups with 12-50 row blocks each, ranges crossing strips
    and images, the DMA ring across raw barriers with counted waits.  Every run equal bit for bit; against the pixel-patch kernel to a
    rounding step; no pixel left unwritten (NaN-filled destination)."""
    Ci, Co = geom
    bf = torch.bfloat16
    N, H, W = (16, 160, 160) if Ci == 256 else (16, 80, 160)
    g = torch.Generator(device=DEV).manual_seed(3)
    for mode in (0, 1):
        w = rnd(Co, Ci, 3, 3, seed=2, scale=0.05) if mode == 0 else rnd(Ci, Co, 3, 3, seed=2, scale=0.05)
        x = torch.randn(N, H, W, Ci, device=DEV, generator=g).to(bf)
        wp = pack_t(w, 0, 1, 1, Ci) if mode == 0 else pack_t(w, 1, 1, 1)
        assert _is_wres(bf, mode, N, H, W, Ci, Co)
        outs = []
        for run in range(3):
            y = torch.full((N, H, W, Co), float('nan'), device=DEV, dtype=bf)
            igemm_t(x, wp, None, y, 3, 1, 1, mode)
            torch.cuda.synchronize()
            outs.append(y)
        assert torch.isfinite(outs[0].float()).all()
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), 'mode %d: runs differ' % mode
        try:
            L().dbn_set_wres16(0)
            y0 = torch.zeros_like(outs[0])
            igemm_t(x, wp, None, y0, 3, 1, 1, mode)
            torch.cuda.synchronize()
        finally:
            L().dbn_set_wres16(1)
        d = (outs[0].float() - y0.float()).abs()
        tol = 4e-5 * float(y0.float().abs().max()) + 1.01 * ULP[bf] * y0.float().abs()
        assert bool((d <= tol).all()), 'mode %d: max excess %g' % (mode, float((d - tol).max()))
