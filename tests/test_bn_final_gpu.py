"""-m gpu: the in-kernel finalize of the train-mode BatchNorm statistics (dbn_conv_bn_set_final; csrc/igemm_common.h dbn_bn_stats_finish):
the conv whose epilogue wrote the per-tile rows also folds them — last workgroup of every 64 rows, then last of the groups — and writes
what bn_finalize_tiles_kernel would (/root/reference/src/modules/resnet.py:73-91, basic.py:32-36: conv -> nn.BatchNorm2d in train mode).
Against the separate finalize kernel on the same rows (equal up to the order of the fp64 merge), on every kernel family that carries it:
the implicit-GEMM tiles (fp32, strided, 1x1, parity-class transposed conv), the Winograd kernel, the pixel-patch and weight-resident
16-bit kernels; counters must come back zero; repeated calls are bit-reproducible."""
import pytest
import torch

from gpu_util import DEV, L, nhwc, report, rnd, stream
from db_text_minimal_amd import _lib
from test_ops_gpu import AT_OF, pack_t

pytestmark = pytest.mark.gpu

# (N, Ci, Co, k, stride, pad, H, W, mode, dtype)
CASES = [(2, 64, 64, 3, 1, 1, 16, 16, 0, torch.float32), (3, 16, 128, 3, 2, 1, 18, 14, 0, torch.float32), (2, 64, 256, 1, 1, 0, 9, 7, 0, torch.float32),
         (16, 64, 64, 3, 1, 1, 80, 80, 0, torch.float32),      # 800 rows of 128 pixels: 13 groups
         (4, 128, 128, 3, 1, 1, 24, 32, 0, torch.bfloat16),    # weight-resident kernel, 8-wave workgroups
         (6, 64, 64, 3, 1, 1, 40, 64, 0, torch.bfloat16),      # weight-resident kernel, 2-wave workgroups, empty rows beyond its grid
         (2, 64, 128, 3, 1, 1, 24, 48, 0, torch.bfloat16),     # pixel-patch kernel
         (2, 128, 64, 3, 2, 1, 20, 20, 0, torch.bfloat16)]     # LDS-DMA ring


@pytest.mark.parametrize('case', CASES)
def test_conv_bn_with_the_statistics_folded_in_kernel(case):
    N, Ci, Co, k, s, p, H, W, mode, dt = case
    at = AT_OF[dt]
    ns = 0 if at == 0 else 1
    kind = 0 if at == 0 else 1
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = nhwc(rnd(N, Ci, H, W, seed=1) * 1.3 + 0.2).to(dt)
    w = rnd(Co, Ci, k, k, seed=2, scale=(2.0 / (Ci * k * k))**0.5)
    bias = rnd(Co, seed=3).to(DEV)
    gam, bet = (rnd(Co, seed=5) * 0.2 + 1).to(DEV), rnd(Co, seed=6).to(DEV)
    wp = pack_t(w, 0, s, kind, Ci)
    rows = L().dbn_igemm_bn_rows(at, ns, N, H, W, Ci, Ho, Wo, Co, k, k, s, p, 0, 0)
    cnt = torch.zeros(L().dbn_igemm_bn_final_counters(rows, Co), device=DEV, dtype=torch.int32)
    grp = torch.full((L().dbn_conv_bn_final_group_doubles(rows, Co), ), float('nan'), device=DEV, dtype=torch.float64)

    def run(fin):
        y = torch.full((N, Ho, Wo, Co), float('nan'), device=DEV, dtype=dt)
        rm_, rv_ = torch.full((Co, ), 0.25, device=DEV), torch.full((Co, ), 2.0, device=DEV)
        sc, sh, mu, rs = (torch.full((Co, ), float('nan'), device=DEV) for _ in range(4))
        ws = torch.full((L().dbn_conv_bn_ws_floats(N, Ho, Wo, Co, 0, s), ), float('nan'), device=DEV)
        if fin:
            _lib.check(L().dbn_conv_bn_set_final(cnt.data_ptr(), grp.data_ptr()), 'set_final')
        _lib.check(L().dbn_conv_bn_t(at, x.data_ptr(), wp.data_ptr(), bias.data_ptr(), y.data_ptr(), N, H, W, Ci, Ho, Wo, Co, k, k, s, p, 0, 0, 0, ns,
                                     gam.data_ptr(), bet.data_ptr(), 1e-5, 0.1, rm_.data_ptr(), rv_.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                     mu.data_ptr(), rs.data_ptr(), ws.data_ptr(), stream()), 'conv_bn_t')
        torch.cuda.synchronize()
        return dict(y=y, scale=sc, shift=sh, mean=mu, rstd=rs, run_mean=rm_, run_var=rv_)

    ref = run(False)
    a, b = run(True), run(True)
    assert int(cnt.abs().sum()) == 0, 'the finalize left a counter behind'
    for key in ref:
        assert torch.isfinite(a[key].float()).all(), key
        assert torch.equal(a[key], b[key]), 'not bit-reproducible: ' + key
    assert torch.equal(a['y'], ref['y'])
    for key in ('scale', 'shift', 'mean', 'rstd', 'run_mean', 'run_var'):
        report('%s: folded in the kernel vs the finalize kernel' % key, a[key].cpu(), ref[key].cpu(), 1e-7, 2e-7)


@pytest.mark.parametrize('shape', [(2, 64, 64, 24, 32), (16, 64, 64, 160, 160), (3, 256, 64, 16, 48)])
def test_winograd_conv_bn_with_the_statistics_folded_in_kernel(shape):
    N, Ci, Co, H, W = shape
    x = nhwc(rnd(N, Ci, H, W, seed=1))
    w = rnd(Co, Ci, 3, 3, seed=2, scale=(2.0 / (Ci * 9))**0.5).to(DEV)
    bias = rnd(Co, seed=3).to(DEV)
    gam, bet = (rnd(Co, seed=5) * 0.2 + 1).to(DEV), rnd(Co, seed=6).to(DEV)
    assert L().dbn_winograd_eligible(N, H, W, Ci, Co)
    up = torch.empty(L().dbn_winograd_panel_floats(Co, Ci), device=DEV)
    _lib.check(L().dbn_winograd_pack(w.data_ptr(), Co, Ci, Ci, 0, up.data_ptr(), stream()), 'winograd_pack')
    rows = L().dbn_winograd_rows(N, H, W)
    cnt = torch.zeros(L().dbn_igemm_bn_final_counters(rows, Co), device=DEV, dtype=torch.int32)
    grp = torch.full((L().dbn_conv_bn_final_group_doubles(rows, Co), ), float('nan'), device=DEV, dtype=torch.float64)

    def run(fin):
        y = torch.full((N, H, W, Co), float('nan'), device=DEV)
        rm_, rv_ = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
        sc, sh, mu, rs = (torch.full((Co, ), float('nan'), device=DEV) for _ in range(4))
        ws = torch.full((L().dbn_winograd_ws_floats(N, H, W, Co), ), float('nan'), device=DEV)
        if fin:
            _lib.check(L().dbn_conv_bn_set_final(cnt.data_ptr(), grp.data_ptr()), 'set_final')
        _lib.check(L().dbn_winograd_conv_bn_f32(x.data_ptr(), up.data_ptr(), bias.data_ptr(), y.data_ptr(), N, H, W, Ci, Co, gam.data_ptr(), bet.data_ptr(),
                                                1e-5, 0.1, rm_.data_ptr(), rv_.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(),
                                                ws.data_ptr(), stream()), 'winograd_conv_bn')
        torch.cuda.synchronize()
        return dict(y=y, scale=sc, shift=sh, mean=mu, rstd=rs, run_mean=rm_, run_var=rv_)

    ref = run(False)
    a, b = run(True), run(True)
    assert int(cnt.abs().sum()) == 0
    for key in ref:
        assert torch.equal(a[key], b[key]), 'not bit-reproducible: ' + key
    assert torch.equal(a['y'], ref['y'])
    for key in ('scale', 'shift', 'mean', 'rstd', 'run_mean', 'run_var'):
        report('%s: folded in the kernel vs the finalize kernel' % key, a[key].cpu(), ref[key].cpu(), 1e-7, 2e-7)


def test_train_step_with_the_statistics_folded_in_kernel_equals_the_default_step():
    """Engine.bn_final_in_kernel (DBN_BN_FINAL=1; off by default: measured slower): one train step of ResNet18-FPN-DBHead with
    every conv + BatchNorm folding its own statistics against the default step (finalize kernels) — losses, maps and gradients to fp64
    merge-order noise."""
    from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
    from oracle import dbnet_oracle as O
    img, gts = O.synthetic_batch(2, 128, seed=31)
    sd = O.new_state(31)
    outs = []
    for on in (False, True):
        model = DBTextModel()
        model.load_state_dict(sd)
        model = model.to(DEV).train()
        model.engine.bn_final_in_kernel = on
        tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
        preds, losses = tr.step(img.to(DEV), gts.to(DEV))
        torch.cuda.synchronize()
        outs.append((preds.clone(), losses.clone(), model.engine.flat_grad.clone(), model.state_dict()['backbone.layer1.0.bn1.running_var'].clone()))
    report('maps', outs[1][0].cpu(), outs[0][0].cpu(), 1e-6, 1e-5)
    report('losses', outs[1][1].cpu(), outs[0][1].cpu(), 1e-6, 1e-6)
    g0, g1 = outs[0][2], outs[1][2]
    assert float((g1 - g0).norm() / g0.norm()) < 1e-5
    report('running variance', outs[1][3].cpu(), outs[0][3].cpu(), 1e-7, 1e-6)
