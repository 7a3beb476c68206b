"""DBLoss — MI355X-native drop-in for /root/reference/src/losses.py:86-139.

`DBLoss(alpha, beta, reduction, negative_ratio, eps)`; `forward(preds, gts)` with
`preds [N,3|2,H,W]`, `gts [4,N,H,W]` (prob_gt, supervision_mask, thresh_gt,
text_area) returns `(prob_loss, threshold_loss, binary_loss, prob_threshold_loss,
total_loss)` for 3-channel preds and the single `prob + beta*thresh` value for
2-channel preds — 0-dim device tensors, differentiable w.r.t. `preds`.

The reductions and the gradient are two HIP kernels (dbn_db_loss_fwd/_bwd).
With the default `reduction='mean'` the reference's "OHEM" term is a scalar BCE
re-weighted by counts; the kernel evaluates that closed form,
which is exact for binary gt/mask maps (what the reference's loader produces).
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


class _DBLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds, gts, alpha, beta, negative_ratio, eps, per_pixel):
        L = _lib.lib()
        N, C, H, W = preds.shape
        st = torch.cuda.current_stream(preds.device).cuda_stream
        losses = torch.empty(5, device=preds.device, dtype=torch.float32)
        coef = torch.zeros(8, device=preds.device, dtype=torch.float32)
        if per_pixel == 1:
            ws = torch.empty(L.dbn_db_loss_ohem_ws_bytes(N, H, W) // 4 + 1, device=preds.device, dtype=torch.float32)
            check(L.dbn_db_loss_ohem_fwd(preds.data_ptr(), gts.data_ptr(), N, H, W, C, alpha, beta, float(negative_ratio), eps,
                                         losses.data_ptr(), coef.data_ptr(), ws.data_ptr(), st), 'db_loss_ohem_fwd')
        else:
            ws = torch.empty(L.dbn_db_loss_ws_bytes() // 4, device=preds.device, dtype=torch.float32)  # (scratch: the library clears its counter)
            fwd = L.dbn_db_loss_sum_fwd if per_pixel == 2 else L.dbn_db_loss_fwd
            check(fwd(preds.data_ptr(), gts.data_ptr(), N, H, W, C, alpha, beta, float(negative_ratio), eps,
                      losses.data_ptr(), coef.data_ptr(), ws.data_ptr(), st), 'db_loss_fwd')
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
        return dpreds, None, None, None, None, None, None


class DBLoss(nn.Module):
    def __init__(self, alpha=1.0, beta=10.0, reduction='mean', negative_ratio=3, eps=1e-6):
        super().__init__()
        if reduction not in ('mean', 'sum', 'none'):  # the strings F.binary_cross_entropy accepts (losses.py:30)
            raise ValueError('%s is not a valid value for reduction' % reduction)
        self.alpha = float(alpha)
        self.beta = float(beta)
        self.reduction = reduction
        self.negative_ratio = negative_ratio
        self.eps = float(eps)

    def forward(self, preds, gts):
        assert preds.dim() == 4
        assert gts.dim() == 4
        if not preds.is_cuda:
            raise RuntimeError('DBLoss runs on MI355X only: preds must be a HIP tensor')
        assert preds.size(1) in (2, 3) and gts.size(0) == 4 and gts.shape[1:] == (preds.size(0), preds.size(2), preds.size(3))
        preds = preds.contiguous().float()
        gts = gts.contiguous().float()
        mode = {'mean': 0, 'none': 1, 'sum': 2}[self.reduction]
        return _DBLossFunction.apply(preds, gts, self.alpha, self.beta, self.negative_ratio, self.eps, mode)
