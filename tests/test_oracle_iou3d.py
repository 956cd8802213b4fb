"""Pins the oracle's restatement of the reference's rotated-BEV overlap / IoU / NMS (oracle/btc_oracle.c, SURVEY.md §8f
row 1; parity unpinned: the reference's CPU twin cannot be built here and it holds no vectors) against an independent
float64 Sutherland-Hodgman clipper, closed-form cases and the defining properties of greedy NMS."""
import numpy as np
import pytest

from oracle import oracle as orc


def _corners(b):
    c, s = np.cos(b[6]), np.sin(b[6])
    loc = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], dtype=np.float64) * np.array([b[3], b[4]]) / 2
    R = np.array([[c, -s], [s, c]])
    return loc @ R.T + np.array([b[0], b[1]], dtype=np.float64)


def _clip_area(pa, pb):
    """area of the intersection of two convex CCW polygons (float64)"""
    out = [tuple(p) for p in pa]
    n = len(pb)
    for i in range(n):
        a, b = pb[i], pb[(i + 1) % n]
        inp, out = out, []
        if not inp:
            break
        side = lambda p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        for j in range(len(inp)):
            p, q = inp[j], inp[(j + 1) % len(inp)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if sp * sq < 0:
                t = sp / (sp - sq)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    if len(out) < 3:
        return 0.0
    x, y = np.array([p[0] for p in out]), np.array([p[1] for p in out])
    return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def rand_boxes(rng, n, spread=12.0):
    b = np.zeros((n, 7), np.float32)
    b[:, :2] = rng.uniform(-spread, spread, (n, 2))
    b[:, 2] = rng.uniform(-1, 1, n)
    b[:, 3] = rng.uniform(1.5, 5.0, n)
    b[:, 4] = rng.uniform(1.0, 2.5, n)
    b[:, 5] = rng.uniform(1.2, 2.0, n)
    b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    return b


def test_overlap_closed_form():
    a = np.array([[0, 0, 0, 4, 2, 1, 0.0], [1, 0, 0, 4, 2, 1, 0.0], [0, 0, 0, 2, 2, 1, np.pi / 4], [10, 10, 0, 1, 1, 1, 0.3],
                  [0, 0, 0, 4, 2, 1, np.pi]], np.float32)
    ov = orc.boxes_overlap_bev(a, a)
    np.testing.assert_allclose(ov[0, 0], 8.0, rtol=1e-6)
    np.testing.assert_allclose(ov[0, 1], 6.0, rtol=1e-6)
    np.testing.assert_allclose(ov[0, 2], 4 - 2 * (np.sqrt(2) - 1) ** 2, rtol=1e-5)   # 45-degree square clipped by the 2 m strip
    assert ov[0, 3] == 0.0 and ov[3, 0] == 0.0
    np.testing.assert_allclose(ov[0, 4], 8.0, rtol=1e-5)                              # heading pi is the same rectangle
    iou = orc.boxes_iou_bev(a, a)
    np.testing.assert_allclose(iou[0, 1], 6.0 / 10.0, rtol=1e-6)
    np.testing.assert_allclose(np.diag(iou), 1.0, atol=1e-5)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_overlap_vs_float64_polygon_clipping(seed):
    rng = np.random.default_rng(seed)
    a, b = rand_boxes(rng, 60, 6.0), rand_boxes(rng, 50, 6.0)
    ov = orc.boxes_overlap_bev(a, b)
    ref = np.array([[_clip_area(_corners(x), _corners(y)) for y in b] for x in a])
    # the reference counts a corner as inside with a 1 cm margin (iou3d_nms_kernel.cu:55-64): up to ~ margin x edge length
    np.testing.assert_allclose(ov, ref, rtol=2e-3, atol=0.06)
    assert (ov > 0).sum() > 200 and (ov == 0).sum() > 200
    np.testing.assert_allclose(orc.boxes_overlap_bev(b, a), ov.T, rtol=1e-4, atol=1e-4)  # symmetric up to fp32 rounding


def test_iou3d_height_term():
    a = np.array([[0, 0, 0, 4, 2, 2, 0.2]], np.float32)
    b = np.array([[0, 0, 1, 4, 2, 2, 0.2], [0, 0, 3, 4, 2, 2, 0.2]], np.float32)
    iou = orc.boxes_iou3d(a, b)
    np.testing.assert_allclose(iou[0, 0], 8.0 / (16 + 16 - 8), rtol=1e-4)   # half the height overlaps
    assert iou[0, 1] == 0.0


@pytest.mark.parametrize("rotated", [True, False])
def test_nms_properties(rotated):
    rng = np.random.default_rng(7)
    boxes = rand_boxes(rng, 300, 10.0)
    scores = rng.uniform(0, 1, 300).astype(np.float32)
    thresh = 0.25
    keep = orc.nms(boxes, scores, thresh, rotated=rotated)
    assert len(set(keep.tolist())) == len(keep) and np.all(np.diff(scores[keep]) <= 0)   # unique, in descending score order
    if not rotated:
        boxes = boxes.copy()
        boxes[:, 6] = 0
    iou = orc.boxes_iou_bev(boxes, boxes)
    kk = iou[np.ix_(keep, keep)]
    assert np.all(kk[np.triu_indices(len(keep), 1)] <= thresh + 1e-6)                     # kept boxes do not overlap above thresh
    dropped = np.setdiff1d(np.arange(300), keep)
    for d in dropped:                                                                    # every dropped box lost to a better kept one
        better = keep[scores[keep] >= scores[d]]
        assert np.any(iou[better, d] > thresh - 1e-6)
    assert orc.nms(boxes, scores, thresh, pre_maxsize=50, rotated=rotated).shape[0] <= 50
    assert orc.nms(boxes[:0], scores[:0], thresh).shape[0] == 0
