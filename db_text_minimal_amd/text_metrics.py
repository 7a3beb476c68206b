"""On-device replacement of the reference's per-step pixel metric.

Reference: `RunningScore` + `cal_text_score` in /root/reference/src/text_metrics.py:9-82, called every
training step (train.py:176-181).  There it costs three device->host copies of [N,H,W] maps
(78.6 MB at bs16) plus numpy thresholding and `np.bincount`; here one HIP kernel accumulates the 2x2
confusion matrix in device memory and only 5 doubles cross PCIe when scores are read.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check


class RunningScore:
    """Same interface as the reference class for n_classes == 2 (`hps.no_classes`)."""

    def __init__(self, n_classes=2):
        if n_classes != 2:
            raise NotImplementedError('the DBNet pixel metric is binary (text / background)')
        self.n_classes = n_classes
        self._hist = None  # device: [unused, n01, n10, n11, total]

    def _dev_hist(self, device):
        if self._hist is None or self._hist.device != device:
            self._hist = torch.zeros(5, device=device, dtype=torch.float64)
        return self._hist

    def update_device(self, prob, gt, mask, thresh):
        """prob: [N,H,W] view of preds[:,0] (or any strided-by-image plane), gt/mask: [N,H,W] contiguous."""
        assert prob.dim() == 3 and prob.is_cuda
        N, H, W = prob.shape
        assert prob.stride(2) == 1 and prob.stride(1) == W, 'probability plane must be row-contiguous'
        gt = gt.contiguous().float()
        mask = mask.contiguous().float()
        h = self._dev_hist(prob.device)
        st = torch.cuda.current_stream(prob.device).cuda_stream
        check(_lib.lib().dbn_pixel_confusion(prob.data_ptr(), prob.stride(0), gt.data_ptr(), mask.data_ptr(), N, H, W, float(thresh),
                                             h.data_ptr(), st), 'pixel_confusion')

    @property
    def confusion_matrix(self):
        if self._hist is None:
            return np.zeros((2, 2))
        _, n01, n10, n11, tot = self._hist.cpu().tolist()
        return np.array([[tot - n01 - n10 - n11, n01], [n10, n11]])

    def get_scores(self):
        """text_metrics.py:36-58, verbatim formulas on the 2x2 matrix."""
        hist = self.confusion_matrix
        acc = np.diag(hist).sum() / (hist.sum() + 0.0001)
        acc_cls = np.diag(hist) / (hist.sum(axis=1) + 0.0001)
        acc_cls = np.nanmean(acc_cls)
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist) + 0.0001)
        mean_iu = np.nanmean(iu)
        freq = hist.sum(axis=1) / (hist.sum() + 0.0001)
        fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
        cls_iu = dict(zip(range(self.n_classes), iu))
        return {'Overall Acc': acc, 'Mean Acc': acc_cls, 'FreqW Acc': fwavacc, 'Mean IoU': mean_iu}, cls_iu

    def reset(self):
        if self._hist is not None:
            self._hist.zero_()


def cal_text_score(texts, gt_texts, training_masks, running_metric_text, thresh=0.5):
    """Drop-in for text_metrics.cal_text_score (text_metrics.py:63-82); `texts` = preds[:, 0, :, :]."""
    running_metric_text.update_device(texts.detach(), gt_texts, training_masks, thresh)
    score_text, _ = running_metric_text.get_scores()
    return score_text
