"""TEST INFRASTRUCTURE ONLY — CPU restatement of the two array operations of the reference's
SegDetectorRepresenter that SURVEY §8(f-3) moves to the device:

  * binarize            src/postprocess.py:51-52      pred > thresh
  * box_score_fast      src/postprocess.py:186-198    mean of the probability map inside a box / polygon

box_score_fast rasterises the polygon with cv2.fillPoly and averages with cv2.mean(bitmap, mask).  OpenCV is a
third-party dependency that is absent from /root/reference and from this image (requirements.txt pins
opencv-python==4.2.0.34), so its rasteriser is restated here from the published algorithm (modules/imgproc/src/drawing.cpp:
fillPoly -> CollectPolyEdges + FillEdgeCollection, Line -> LineIterator, XY_SHIFT = 16):
  mask = the 8-connected Bresenham line of every polygon edge (integer vertices)
       U the even-odd scanline fill: on scanline y every non-horizontal edge with y0 <= y < y1 contributes
         x = x0 + (y - y0) * dx in 16.16 fixed point (dx = ((x1 - x0) << 16) / (y1 - y0), truncating); sorted
         crossings are paired and pixels ceil(xa) .. floor(xb) are set.
PARITY UNPINNED: neither the reference nor this image can run cv2, and the reference holds no fixture for this
function; the restatement is pinned only by analytic cases (axis-aligned rectangles, triangles, degenerate boxes)
in tests/test_postprocess_cpu.py.  Only tests/ may import this module.
"""
import numpy as np

XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT


def binarize(pred, thresh=0.3):
    """postprocess.py:51-52."""
    return pred > thresh


def line_pixels(p1, p2):
    """Pixels of cv2.line(img, p1, p2, color, 8) — LineIterator(connectivity 8, leftToRight=True), literal."""
    (x1, y1), (x2, y2) = (int(p1[0]), int(p1[1])), (int(p2[0]), int(p2[1]))
    dx, dy = x2 - x1, y2 - y1
    if dx < 0:  # walk left to right
        x1, y1, dx, dy = x2, y2, -dx, -dy
    sy = -1 if dy < 0 else 1
    dy = abs(dy)
    steep = dy > dx
    if steep:
        dx, dy = dy, dx
    err = dx - 2 * dy
    x, y = x1, y1
    out = []
    for _ in range(dx + 1):
        out.append((x, y))
        minor = err < 0
        err += -2 * dy + (2 * dx if minor else 0)
        if steep:
            y += sy
            x += 1 if minor else 0
        else:
            x += 1
            y += sy if minor else 0
    return out


def fill_poly_mask(h, w, pts):
    """mask (h, w) uint8 of cv2.fillPoly(mask, [pts], 1) for integer vertices pts (P, 2) = (x, y)."""
    pts = np.asarray(pts, dtype=np.int64).reshape(-1, 2)
    mask = np.zeros((h, w), np.uint8)
    n = len(pts)
    edges = []
    for i in range(n):
        (x0, y0), (x1, y1) = pts[i - 1], pts[i]
        for (x, y) in line_pixels((x0, y0), (x1, y1)):
            if 0 <= x < w and 0 <= y < h:
                mask[y, x] = 1
        if y0 == y1:
            continue
        if y0 > y1:
            x0, y0, x1, y1 = x1, y1, x0, y0
        num, den = (int(x1) - int(x0)) << XY_SHIFT, int(y1) - int(y0)
        dxf = abs(num) // den * (1 if num >= 0 else -1)  # C++ integer division truncates toward zero
        edges.append((int(y0), int(y1), int(x0) << XY_SHIFT, dxf))
    if not edges:
        return mask
    for y in range(max(0, min(e[0] for e in edges)), min(h, max(e[1] for e in edges))):
        xs = sorted(x0 + (y - y0) * dxf for (y0, y1, x0, dxf) in edges if y0 <= y < y1)
        for a, b in zip(xs[0::2], xs[1::2]):
            xa, xb = (a + XY_ONE - 1) >> XY_SHIFT, b >> XY_SHIFT
            if xa < w and xb >= 0:
                mask[y, max(xa, 0):min(xb, w - 1) + 1] = 1
    return mask


def box_score_fast(bitmap, _box):
    """postprocess.py:186-198 (np.int of the original == int)."""
    h, w = bitmap.shape[:2]
    box = np.array(_box, dtype=np.float32).reshape(-1, 2).copy()
    xmin = int(np.clip(np.floor(box[:, 0].min()).astype(int), 0, w - 1))
    xmax = int(np.clip(np.ceil(box[:, 0].max()).astype(int), 0, w - 1))
    ymin = int(np.clip(np.floor(box[:, 1].min()).astype(int), 0, h - 1))
    ymax = int(np.clip(np.ceil(box[:, 1].max()).astype(int), 0, h - 1))
    box[:, 0] = box[:, 0] - xmin
    box[:, 1] = box[:, 1] - ymin
    mask = fill_poly_mask(ymax - ymin + 1, xmax - xmin + 1, box.astype(np.int32))
    cnt = int(mask.sum())
    if cnt == 0:  # cv2.mean over an empty mask returns 0
        return 0.0
    region = bitmap[ymin:ymax + 1, xmin:xmax + 1].astype(np.float64)
    return float((region * mask).sum() / cnt)
