"""Helpers shared by the -m gpu parity tests (call the C ABI through ctypes)."""
import numpy as np
import torch

from db_text_minimal_amd import _lib

DEV = 'cuda'


def L():
    return _lib.lib()


def stream():
    return torch.cuda.current_stream().cuda_stream


def nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2).contiguous().cpu()


def report(tag, got, ref, atol, rtol):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (tag, got.shape, ref.shape)
    assert torch.isfinite(got).all(), tag + ': non-finite values'
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    worst = float((err / tol.clamp_min(1e-300)).max()) if err.numel() else 0.0
    if float(err.max() if err.numel() else 0) == 0.0:
        worst = 0.0
    msg = '%s: max abs err %.3e (ref max %.3e), worst err/tol %.3f' % (tag, float(err.max()) if err.numel() else 0,
                                                                      float(ref.abs().max()) if ref.numel() else 0, worst)
    print(msg)
    assert worst <= 1.0, msg


def pack(w, mode, stride=1, ns=0):
    """w: CPU OIHW tensor -> device panels (ns = 0 fp32 MFMA, 3 = bf16x3 split, 1 = bf16)."""
    O, I, R, S = w.shape
    wd = w.contiguous().to(DEV)
    if ns == 0:
        out = torch.empty(L().dbn_igemm_panel_floats(O, I, R, S, mode, stride), device=DEV)
        _lib.check(L().dbn_pack_weights(wd.data_ptr(), O, I, R, S, mode, stride, out.data_ptr(), stream()), 'pack')
    else:
        out = torch.empty(L().dbn_igemm_bf16s_panel_floats(O, I, R, S, mode, stride, ns), device=DEV)
        _lib.check(L().dbn_pack_weights_bf16s(wd.data_ptr(), O, I, R, S, mode, stride, ns, out.data_ptr(), stream()), 'pack')
    return out


def igemm(src, wpk, bias, dst, R, stride, pad, mode, accumulate=0, tile=0, ns=0):
    N, Hs, Ws, Cs = src.shape
    _, Hd, Wd, Cd = dst.shape
    args = (src.data_ptr(), wpk.data_ptr(), None if bias is None else bias.data_ptr(), dst.data_ptr(), N, Hs, Ws, Cs, Hd, Wd, Cd, R,
            R, stride, pad, mode, accumulate, tile)
    if ns == 0:
        _lib.check(L().dbn_igemm_f32(*args, stream()), 'igemm')
    else:
        _lib.check(L().dbn_igemm_bf16s(*args, ns, stream()), 'igemm_bf16s')


def wgrad(sm, big, O, I, k, stride, pad, scale=1.0, ns=0):
    N, Ho, Wo, _ = sm.shape
    _, H, W, Cb = big.shape
    slab = torch.empty(L().dbn_wgrad_slab_floats_hw(N, Ho, Wo, O, H, W, Cb, k, k, 4), device=DEV)
    g = torch.full((O, I, k, k), float('nan'), device=DEV)
    args = (sm.data_ptr(), big.data_ptr(), slab.data_ptr(), g.data_ptr(), N, Ho, Wo, O, H, W, Cb, I, k, k, stride, pad, scale)
    if ns == 0:
        _lib.check(L().dbn_wgrad_f32(*args, stream()), 'wgrad')
    else:
        _lib.check(L().dbn_wgrad_bf16s(*args, ns, stream()), 'wgrad_bf16s')
    return g


def reduce_ws():
    return torch.empty(L().dbn_reduce_ws_floats(512), device=DEV)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def report_robust(tag, got, ref, atol, rtol, frac=0.999):
    """Like report(), but tolerates a tiny fraction of outliers (Adam turns a sign flip of a
    ~zero gradient into a +-lr step, so a handful of elements legitimately differ)."""
    got = got.detach().cpu().double().reshape(-1)
    ref = ref.detach().cpu().double().reshape(-1)
    assert got.shape == ref.shape and torch.isfinite(got).all(), tag
    ok = (got - ref).abs() <= atol + rtol * ref.abs()
    f = float(ok.double().mean())
    print('%s: %.5f of elements within tol, mean abs err %.3e' % (tag, f, float((got - ref).abs().mean())))
    assert f >= frac, tag
