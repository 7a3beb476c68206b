"""-m gpu: on-device pixel metric vs the reference's cal_text_score output (golden)."""
import os

import numpy as np
import pytest
import torch

from gpu_util import DEV
from db_text_minimal_amd.text_metrics import RunningScore, cal_text_score
from oracle import dbnet_oracle as O

pytestmark = pytest.mark.gpu


def test_pixel_metric_matches_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, 'pixel_metrics.npz'))
    rs = RunningScore(2)
    for step in range(2):
        P, G, M = (torch.from_numpy(z['step%d/%s' % (step, k)]).to(DEV) for k in 'PGM')
        preds = torch.stack([P, torch.zeros_like(P), torch.zeros_like(P)], 1).contiguous()  # [N,3,H,W] like the model output
        score = cal_text_score(preds[:, 0, :, :], G, M, rs, thresh=0.25)
        assert np.array_equal(rs.confusion_matrix, z['step%d/hist' % step])  # integer counts: bit-exact
        got = [score[k] for k in ('Overall Acc', 'Mean Acc', 'FreqW Acc', 'Mean IoU')]
        assert np.allclose(got, z['step%d/scores' % step], rtol=1e-12)
    rs.reset()
    assert rs.confusion_matrix.sum() == 0


def test_pixel_metric_full_size_and_edges():
    g = torch.Generator(device=DEV).manual_seed(0)
    N, S = 16, 640
    P = torch.rand(N, 3, S, S, device=DEV, generator=g)
    G = (torch.rand(N, S, S, device=DEV, generator=g) > 0.9).float()
    M = (torch.rand(N, S, S, device=DEV, generator=g) > 0.05).float()
    rs = RunningScore(2)
    cal_text_score(P[:, 0], G, M, rs, thresh=0.5)
    hist = rs.confusion_matrix
    assert hist.sum() == N * S * S  # every pixel is counted exactly once
    ref = O.pixel_confusion(P[:2, 0], G[:2], M[:2], 0.5)
    rs2 = RunningScore(2)
    cal_text_score(P[:2, 0], G[:2], M[:2], rs2, thresh=0.5)
    assert np.array_equal(rs2.confusion_matrix, ref)
    # all masked out -> everything lands in hist[0][0]; threshold is strict (> thresh)
    rs3 = RunningScore(2)
    cal_text_score(torch.full((1, 8, 8), 0.5, device=DEV), torch.ones(1, 8, 8, device=DEV), torch.ones(1, 8, 8, device=DEV), rs3, 0.5)
    assert np.array_equal(rs3.confusion_matrix, np.array([[0, 0], [64, 0]]))
    cal_text_score(torch.ones(1, 8, 8, device=DEV), torch.ones(1, 8, 8, device=DEV), torch.zeros(1, 8, 8, device=DEV), rs3, 0.5)
    assert np.array_equal(rs3.confusion_matrix, np.array([[64, 0], [64, 0]]))
