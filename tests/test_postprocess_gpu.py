"""-m gpu: device binarise / box scores vs the restated reference functions."""
import numpy as np
import pytest
import torch

from db_text_minimal_amd.postprocess import binarize_u8, box_scores
from oracle import postprocess_oracle as P

pytestmark = pytest.mark.gpu


def test_binarize_u8_matches_reference_threshold():
    g = torch.Generator().manual_seed(0)
    preds = torch.rand(3, 2, 64, 96, generator=g)
    preds[0, 0, 0, :4] = torch.tensor([0.3, 0.30000001, 0.29999998, 1.0])
    out = binarize_u8(preds.cuda(), 0.3).cpu().numpy()
    ref = P.binarize(preds[:, 0].numpy(), np.float32(0.3)).astype(np.uint8)
    assert out.dtype == np.uint8 and out.shape == (3, 64, 96) and (out == ref).all()


@pytest.mark.parametrize('npts', [4, 7, 12])
def test_box_scores_match_restated_box_score_fast(npts):
    rng = np.random.default_rng(npts)
    H, W = 96, 128
    bm = rng.random((H, W)).astype(np.float32)
    boxes = []
    for k in range(60):
        c = rng.uniform([-10, -10], [W + 10, H + 10])
        if npts == 4:  # rotated rectangles like get_mini_boxes
            a, (hw, hh) = rng.uniform(0, np.pi), rng.uniform(0.3, 25, 2)
            R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
            pts = c + np.array([[-hw, -hh], [hw, -hh], [hw, hh], [-hw, hh]]) @ R.T
        else:  # star-shaped (possibly concave) polygons like approxPolyDP output
            ang = np.sort(rng.uniform(0, 2 * np.pi, npts))
            rad = rng.uniform(2, 30, npts)
            pts = c + np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1)
        boxes.append(pts)
    boxes.append(np.full((npts, 2), 17.5))  # degenerate: a single pixel
    boxes.append(np.array([[-50.0, -50.0]] * npts))  # entirely outside: clipped to the corner pixel
    boxes = np.stack(boxes).astype(np.float32)
    got = box_scores(torch.from_numpy(bm).cuda(), boxes)
    ref = np.array([P.box_score_fast(bm, b) for b in boxes], np.float32)
    err = np.abs(got - ref)
    print('box scores: max abs err %.3e over %d boxes' % (err.max(), len(boxes)))
    assert err.max() < 1e-6


def test_box_scores_on_model_sized_map():
    rng = np.random.default_rng(5)
    bm = rng.random((1280, 1280)).astype(np.float32)
    boxes = np.array([[[100.5, 200.25], [900.75, 220.5], [890.0, 700.0], [90.0, 680.0]],
                      [[0, 0], [1279, 0], [1279, 1279], [0, 1279]]], np.float32)
    got = box_scores(torch.from_numpy(bm).cuda(), boxes)
    ref = np.array([P.box_score_fast(bm, b) for b in boxes], np.float32)
    assert np.abs(got - ref).max() < 1e-6
    assert box_scores(torch.from_numpy(bm).cuda(), np.zeros((0, 4, 2), np.float32)).shape == (0, )
