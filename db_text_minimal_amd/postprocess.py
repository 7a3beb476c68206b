"""Device-side array work of the reference's SegDetectorRepresenter (src/postprocess.py), SURVEY §8(f-3).

The contour tracing itself (cv2.findContours / minAreaRect / pyclipper unclip) stays on the host, unchanged.  What
moves to the GPU are the two operations that touch whole probability maps:

  binarize_u8(preds, thresh)        postprocess.py:51-52   `pred[:, 0] > thresh` as uint8 — the bitmap findContours needs
                                                           crosses PCIe at 1 B/px instead of the 4 B/px float map
  box_scores(prob_map, boxes)       postprocess.py:186-198 box_score_fast for all candidate boxes of an image in one
                                                           launch — the float map never has to leave the device

A maintainer's change in SegDetectorRepresenter.boxes_from_bitmap (postprocess.py:105-141) is two lines: collect the
candidate boxes first, then `scores = box_scores(pred, np.stack(boxes))` instead of calling self.box_score_fast per box.
"""
import numpy as np
import torch

from ._lib import check, lib


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def binarize_u8(preds, thresh=0.3):
    """preds: [N, C, H, W] fp32 device tensor (the DBTextModel output) -> uint8 [N, H, W] bitmap of channel 0 > thresh."""
    assert preds.is_cuda and preds.dtype == torch.float32 and preds.dim() == 4 and preds.is_contiguous()
    N, C, H, W = preds.shape
    out = torch.empty((N, H, W), device=preds.device, dtype=torch.uint8)
    check(lib().dbn_binarize_u8(preds.data_ptr(), N, C, H, W, float(thresh), out.data_ptr(), _stream(preds)), 'binarize_u8')
    return out


def box_scores(prob_map, boxes):
    """prob_map: [H, W] fp32 device tensor (pred[n, 0]); boxes: array-like [K, P, 2] of (x, y) vertices, P <= 64
    -> numpy float32 [K] = box_score_fast(prob_map, boxes[k]) of the reference."""
    assert prob_map.is_cuda and prob_map.dtype == torch.float32 and prob_map.dim() == 2 and prob_map.is_contiguous()
    b = torch.as_tensor(np.ascontiguousarray(np.asarray(boxes, dtype=np.float32)))
    assert b.dim() == 3 and b.shape[2] == 2 and 1 <= b.shape[1] <= 64, b.shape
    K, P = int(b.shape[0]), int(b.shape[1])
    if K == 0:
        return np.zeros((0, ), np.float32)
    bd = b.to(prob_map.device)
    scores = torch.empty(K, device=prob_map.device, dtype=torch.float32)
    H, W = prob_map.shape
    check(lib().dbn_box_scores(prob_map.data_ptr(), H, W, bd.data_ptr(), K, P, scores.data_ptr(), _stream(prob_map)), 'box_scores')
    return scores.cpu().numpy()
