"""-m "not gpu": analytic pins of the restated cv2.fillPoly / box_score_fast (oracle/postprocess_oracle.py)."""
import numpy as np

from oracle import postprocess_oracle as P


def test_line_iterator_is_symmetric_and_8_connected():
    rng = np.random.default_rng(0)
    for _ in range(200):
        a, b = rng.integers(-5, 30, 2), rng.integers(-5, 30, 2)
        pa, pb = P.line_pixels(a, b), P.line_pixels(b, a)
        assert set(pa) == set(pb)  # leftToRight canonicalises the direction
        assert len(pa) == max(abs(int(a[0] - b[0])), abs(int(a[1] - b[1]))) + 1
        for (x0, y0), (x1, y1) in zip(pa[:-1], pa[1:]):
            assert max(abs(x1 - x0), abs(y1 - y0)) == 1
        assert tuple(int(v) for v in a) in pa and tuple(int(v) for v in b) in pa


def test_fill_rectangle_and_triangle():
    m = P.fill_poly_mask(8, 10, [(1, 1), (6, 1), (6, 5), (1, 5)])
    ref = np.zeros((8, 10), np.uint8)
    ref[1:6, 1:7] = 1  # closed rectangle: boundary pixels included
    assert (m == ref).all()
    t = P.fill_poly_mask(12, 12, [(0, 0), (10, 0), (0, 10)])
    for y in range(12):
        for x in range(12):
            assert t[y, x] == (1 if x + y <= 10 and x <= 10 and y <= 10 else 0), (x, y)
    assert P.fill_poly_mask(5, 5, [(2, 2), (2, 2), (2, 2), (2, 2)]).sum() == 1  # degenerate box: one pixel
    clipped = P.fill_poly_mask(4, 4, [(-3, -3), (8, -3), (8, 8), (-3, 8)])
    assert clipped.all()


def test_box_score_fast_rectangle_mean():
    rng = np.random.default_rng(1)
    bm = rng.random((40, 50)).astype(np.float32)
    s = P.box_score_fast(bm, np.array([[10.2, 5.7], [30.9, 5.7], [30.9, 20.1], [10.2, 20.1]], np.float32))
    # xmin=10, ymin=5; vertices truncate to (0,0),(20,0),(20,15),(0,15) -> pixels x 10..30, y 5..20
    assert abs(s - float(bm[5:21, 10:31].astype(np.float64).mean())) < 1e-9
    assert P.box_score_fast(np.zeros((8, 8), np.float32), np.array([[1, 1], [5, 1], [5, 5], [1, 5]], np.float32)) == 0.0
    assert (P.binarize(np.array([0.2, 0.3, 0.31]), 0.3) == np.array([False, False, True])).all()
