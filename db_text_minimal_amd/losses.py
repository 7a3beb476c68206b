"""DBLoss — MI355X-native drop-in for /root/reference/src/losses.py:86-139.

`DBLoss(alpha, beta, reduction, negative_ratio, eps)`; `forward(preds, gts)` with
`preds [N,3|2,H,W]`, `gts [4,N,H,W]` (prob_gt, supervision_mask, thresh_gt,
text_area) returns `(prob_loss, threshold_loss, binary_loss, prob_threshold_loss,
total_loss)` for 3-channel preds and the single `prob + beta*thresh` value for
2-channel preds — 0-dim device tensors, differentiable w.r.t. `preds`.

The reductions and the gradient are two HIP kernels (dbn_db_loss_fwd/_bwd).
With the default `reduction='mean'` the reference's "OHEM" term is a scalar BCE
re-weighted by counts; the kernel evaluates that closed form,
which is exact for binary gt/mask maps (what the reference's loader produces,
data_loaders.py:112-134) — and ONLY for those: the kernel counts the pixels whose
prob_gt or supervision_mask is neither 0 nor 1, and a DBLoss that has seen such a
batch raises ValueError at its next call (or at `check_maps()`; the count comes back
through pinned memory, no host synchronisation on the hot path).  For such maps
construct `DBLoss(..., fractional_maps=True)`: losses.py:33-39 evaluated literally,
`bce * (sum(positive) + topk(negative, n_neg).sum()) / (n_pos + n_neg + eps)`, the
top-k sum by the device radix select (dbn_db_loss_frac_fwd).
`reduction='sum'` sums that scalar instead of averaging it (dbn_db_loss_sum_fwd).
`reduction='none'` is the paper's per-pixel OHEM: the `n_neg` hardest negatives
are selected on device by a 3-pass radix select (dbn_db_loss_ohem_fwd/_bwd)
instead of `torch.topk` over 6.5 M elements.
No host synchronisation happens here (the reference's `int(tensor)` / `assert`
syncs, losses.py:25-27,65, are folded into the finalize kernel).
"""
import torch
from torch import nn

from . import _lib
from ._lib import check


class BinaryMapGuard:
    """Deferred refusal of non-binary prob_gt / supervision_mask maps in the closed-form reductions.  The forward kernel
    leaves the number of offending pixels in coef[7]; `watch` starts an asynchronous copy of it into pinned memory behind
    the kernel, `check` (called at the start of the NEXT loss evaluation, and by DBLoss.check_maps) looks at copies that have
    landed.  The hot path never waits for the device."""

    MESSAGE = ('DBLoss: %d pixels of prob_gt / supervision_mask in an earlier batch were neither 0 nor 1.  With reduction=%r the '
               'OHEM term is evaluated in closed form, exact for binary maps only (reference losses.py:33-39 takes topk over '
               'loss * negative); construct DBLoss(..., fractional_maps=True) to evaluate it literally for such maps.')

    def __init__(self, reduction):
        self.reduction = reduction
        self.host = None
        self.event = None
        self.in_flight = False

    def check(self, wait=False):
        if not self.in_flight or torch.cuda.is_current_stream_capturing():
            return
        if wait:
            self.event.synchronize()
        elif not self.event.query():
            return
        self.in_flight = False
        n = float(self.host[0])
        if n != 0.0:
            raise ValueError(self.MESSAGE % (int(n), self.reduction))

    def watch(self, coef):
        if self.in_flight or torch.cuda.is_current_stream_capturing():
            return  # (the previous copy has not landed yet: this call goes unwatched, the next one is looked at again)
        if self.host is None:
            self.host = torch.zeros(1, dtype=torch.float32).pin_memory()
            self.event = torch.cuda.Event()
        self.host.copy_(coef[7:8], non_blocking=True)
        self.event.record()
        self.in_flight = True


class _DBLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds, gts, alpha, beta, negative_ratio, eps, per_pixel, guard=None):
        L = _lib.lib()
        N, C, H, W = preds.shape
        st = torch.cuda.current_stream(preds.device).cuda_stream
        losses = torch.empty(5, device=preds.device, dtype=torch.float32)
        coef = torch.zeros(8, device=preds.device, dtype=torch.float32)
        if per_pixel == 1:
            ws = torch.empty(L.dbn_db_loss_ohem_ws_bytes(N, H, W) // 4 + 1, device=preds.device, dtype=torch.float32)
            check(L.dbn_db_loss_ohem_fwd(preds.data_ptr(), gts.data_ptr(), N, H, W, C, alpha, beta, float(negative_ratio), eps,
                                         losses.data_ptr(), coef.data_ptr(), ws.data_ptr(), st), 'db_loss_ohem_fwd')
        elif per_pixel in (3, 4):  # 'mean' / 'sum' on non-binary maps: the literal top-k form
            ws = torch.empty(L.dbn_db_loss_ohem_ws_bytes(N, H, W) // 4 + 1, device=preds.device, dtype=torch.float32)
            check(L.dbn_db_loss_frac_fwd(preds.data_ptr(), gts.data_ptr(), N, H, W, C, alpha, beta, float(negative_ratio), eps,
                                         1 if per_pixel == 4 else 0, losses.data_ptr(), coef.data_ptr(), ws.data_ptr(), st), 'db_loss_frac_fwd')
        else:
            ws = torch.empty(L.dbn_db_loss_ws_bytes() // 4, device=preds.device, dtype=torch.float32)  # (scratch: the library clears its counter)
            fwd = L.dbn_db_loss_sum_fwd if per_pixel == 2 else L.dbn_db_loss_fwd
            check(fwd(preds.data_ptr(), gts.data_ptr(), N, H, W, C, alpha, beta, float(negative_ratio), eps,
                      losses.data_ptr(), coef.data_ptr(), ws.data_ptr(), st), 'db_loss_fwd')
        if guard is not None and per_pixel in (0, 2):
            guard.watch(coef)
        ctx.save_for_backward(preds, gts, coef, ws)
        ctx.cfg = (alpha, beta, per_pixel)
        if C == 3:
            return tuple(losses[i] for i in range(5))
        return losses[4]

    @staticmethod
    def backward(ctx, *gouts):
        preds, gts, coef, ws = ctx.saved_tensors
        alpha, beta, per_pixel = ctx.cfg
        L = _lib.lib()
        N, C, H, W = preds.shape
        st = torch.cuda.current_stream(preds.device).cuda_stream
        g = torch.zeros(5, device=preds.device, dtype=torch.float32)
        if C == 3:
            for i, go in enumerate(gouts):
                if go is not None:
                    g[i] = go
        else:
            g[4] = gouts[0]
        dpreds = torch.empty_like(preds)
        if per_pixel == 1:
            check(L.dbn_db_loss_ohem_bwd(preds.data_ptr(), gts.data_ptr(), coef.data_ptr(), g.data_ptr(), ws.data_ptr(), alpha, beta,
                                         N, H, W, C, dpreds.data_ptr(), st), 'db_loss_ohem_bwd')
        else:
            check(L.dbn_db_loss_bwd(preds.data_ptr(), gts.data_ptr(), coef.data_ptr(), g.data_ptr(), alpha, beta, N, H, W, C,
                                    dpreds.data_ptr(), st), 'db_loss_bwd')
        return dpreds, None, None, None, None, None, None, None


class DBLoss(nn.Module):
    def __init__(self, alpha=1.0, beta=10.0, reduction='mean', negative_ratio=3, eps=1e-6, fractional_maps=False):
        """The reference's signature (losses.py:86-91) plus `fractional_maps` (see the module docstring): False = the closed form
        of the scalar-BCE reductions, non-binary maps are refused; True = the literal top-k form for any maps."""
        super().__init__()
        if reduction not in ('mean', 'sum', 'none'):  # the strings F.binary_cross_entropy accepts (losses.py:30)
            raise ValueError('%s is not a valid value for reduction' % reduction)
        self.alpha = float(alpha)
        self.beta = float(beta)
        self.reduction = reduction
        self.negative_ratio = negative_ratio
        self.eps = float(eps)
        self.fractional_maps = bool(fractional_maps)
        self._guard = BinaryMapGuard(reduction)

    def mode(self):
        """Kernel selector: 0 'mean', 1 'none' (per-pixel OHEM), 2 'sum', 3 / 4 'mean' / 'sum' in the literal top-k form."""
        m = {'mean': 0, 'none': 1, 'sum': 2}[self.reduction]
        return m + 3 - (m >> 1) if (self.fractional_maps and m != 1) else m

    def check_maps(self):
        """Waits for the pending non-binary-map count of the last watched call (if any) and raises ValueError on a non-zero one."""
        self._guard.check(wait=True)

    def forward(self, preds, gts):
        assert preds.dim() == 4
        assert gts.dim() == 4
        if not preds.is_cuda:
            raise RuntimeError('DBLoss runs on MI355X only: preds must be a HIP tensor')
        assert preds.size(1) in (2, 3) and gts.size(0) == 4 and gts.shape[1:] == (preds.size(0), preds.size(2), preds.size(3))
        preds = preds.contiguous().float()
        gts = gts.contiguous().float()
        self._guard.check()
        return _DBLossFunction.apply(preds, gts, self.alpha, self.beta, self.negative_ratio, self.eps, self.mode(), self._guard)
