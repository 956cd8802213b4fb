"""rotated BEV overlap / IoU / NMS kernels against the oracle (oracle/btc_oracle.c, pinned in
tests/test_oracle_iou3d.py).  IoU values: same fp32 formulation, tolerance for the last-ulp differences of the device
cosf / sinf / atan2f; NMS: identical kept sets except where an IoU sits within that tolerance of the threshold."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from test_oracle_iou3d import rand_boxes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("na,nb,seed", [(1, 1, 0), (37, 53, 1), (200, 130, 2), (0, 5, 3)])
def test_pairwise_overlap_and_iou(na, nb, seed):
    from btcdet_amd import iou3d_nms
    rng = np.random.default_rng(seed)
    a, b = rand_boxes(rng, na, 6.0), rand_boxes(rng, nb, 6.0)
    iou = iou3d_nms.boxes_iou_bev(_t(a), _t(b)).cpu().numpy()
    np.testing.assert_allclose(iou, orc.boxes_iou_bev(a, b), rtol=1e-4, atol=2e-5)
    iou3 = iou3d_nms.boxes_iou3d_gpu(_t(a), _t(b)).cpu().numpy()
    np.testing.assert_allclose(iou3, orc.boxes_iou3d(a, b), rtol=1e-4, atol=2e-5)
    if na and nb:
        assert iou.shape == (na, nb) and float(iou.max()) <= 1.0 + 1e-5


def test_known_cases_and_degenerate_boxes():
    from btcdet_amd import iou3d_nms
    a = np.array([[0, 0, 0, 4, 2, 1, 0.0], [1, 0, 0, 4, 2, 1, 0.0], [0, 0, 0, 2, 2, 1, np.pi / 4], [10, 10, 0, 1, 1, 1, 0.3],
                  [0, 0, 0, 0, 0, 0, 0.0]], np.float32)
    iou = iou3d_nms.boxes_iou_bev(_t(a), _t(a)).cpu().numpy()
    np.testing.assert_allclose(iou[0, 1], 0.6, rtol=1e-5)
    np.testing.assert_allclose(iou[0, 2], (4 - 2 * (np.sqrt(2) - 1) ** 2) / (8 + 4 - (4 - 2 * (np.sqrt(2) - 1) ** 2)), rtol=1e-4)
    assert iou[0, 3] == 0 and np.all(np.isfinite(iou))
    np.testing.assert_allclose(iou, orc.boxes_iou_bev(a, a), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("n,spread,rotated", [(1, 5.0, True), (64, 6.0, True), (65, 6.0, False), (300, 10.0, True), (1500, 14.0, True),
                                               (1500, 14.0, False), (5000, 30.0, True)])
def test_nms_matches_oracle(n, spread, rotated):
    from btcdet_amd import iou3d_nms
    rng = np.random.default_rng(n)
    boxes = rand_boxes(rng, n, spread)
    scores = rng.permutation(n).astype(np.float32) / n          # distinct scores: the order is unambiguous
    thresh = 0.1 if n >= 1500 else 0.3
    fn = iou3d_nms.nms_gpu if rotated else iou3d_nms.nms_normal_gpu
    keep, _ = fn(_t(boxes), _t(scores), thresh)
    keep = keep.cpu().numpy()
    ref = orc.nms(boxes, scores, thresh, rotated=rotated)
    if not np.array_equal(keep, ref):
        # a decision may flip only where an IoU is within fp32 rounding of the threshold
        bb = boxes.copy()
        if not rotated:
            bb[:, 6] = 0
        iou = orc.boxes_iou_bev(bb, bb)
        assert np.any(np.abs(iou - thresh) < 1e-5), "kept sets differ without an IoU at the threshold"
    else:
        assert np.all(np.diff(scores[keep]) < 0)
    k2, _ = iou3d_nms.nms_gpu(_t(boxes), _t(scores), thresh, pre_maxsize=max(n // 2, 1))
    assert k2.shape[0] <= max(n // 2, 1)


def test_compiled_module_stand_in_signatures():
    """the reference calls the compiled module with preallocated outputs and a CPU int64 `keep` (iou3d_nms_utils.py:27,43,65,93)"""
    from btcdet_amd import iou3d_nms
    m = iou3d_nms.iou3d_nms_cuda
    rng = np.random.default_rng(11)
    a, b = rand_boxes(rng, 40, 5.0), rand_boxes(rng, 30, 5.0)
    out = torch.zeros((40, 30), device=DEV)
    assert m.boxes_iou_bev_gpu(_t(a), _t(b), out) == 1
    np.testing.assert_allclose(out.cpu().numpy(), orc.boxes_iou_bev(a, b), rtol=1e-4, atol=2e-5)
    ov = torch.zeros((40, 30), device=DEV)
    m.boxes_overlap_bev_gpu(_t(a), _t(b), ov)
    np.testing.assert_allclose(ov.cpu().numpy(), orc.boxes_overlap_bev(a, b), rtol=1e-4, atol=1e-4)
    scores = rng.permutation(40).astype(np.float32)
    order = np.argsort(-scores, kind="stable")
    keep = torch.zeros(40, dtype=torch.int64)
    n = m.nms_gpu(_t(a[order]), keep, 0.2)
    np.testing.assert_array_equal(order[keep[:n].numpy()], orc.nms(a, scores, 0.2))
    cpu_out = torch.zeros((40, 30))
    m.boxes_iou_bev_cpu(torch.from_numpy(a), torch.from_numpy(b), cpu_out)
    np.testing.assert_allclose(cpu_out.numpy(), orc.boxes_iou_bev(a, b), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("n,spread,rotated,max_keep", [(100, 4.0, True, 16), (3000, 12.0, True, 256), (9000, 22.0, True, 256), (9000, 22.0, False, 512),
                                                        (2560, 9.0, True, 64), (6000, 6.0, True, 512), (5, 3.0, True, 256)])
def test_nms_topk_is_the_truncated_full_chain(n, spread, rotated, max_keep):
    """btc_nms_topk (the chain stopped at max_keep, chunked, batched, no read-back) == the first max_keep entries of btc_nms's keep list,
    bit for bit (same IoU function, same argument order), for a batch of two different scenes; dense clusters so that suppression
    really happens, sizes on both sides of the 2560-box chunk"""
    from btcdet_amd import iou3d_nms
    rng = np.random.default_rng(n + max_keep)
    scenes = []
    for s in range(2):
        b = rand_boxes(rng, n, spread)
        scenes.append(_t(b))
    thresh = 0.25
    keep, cnt = iou3d_nms.nms_topk(torch.stack(scenes), thresh, max_keep, rotated=rotated)
    assert tuple(keep.shape) == (2, max_keep) and keep.dtype == torch.int64 and cnt.dtype == torch.int32
    for s in range(2):
        full, m = iou3d_nms._nms_sorted(scenes[s], thresh, rotated)
        want = full[:min(m, max_keep)].cpu().numpy()
        got, c = keep[s].cpu().numpy(), int(cnt[s])
        assert c == len(want), (c, len(want), m)
        assert np.array_equal(got[:c], want) and np.all(got[c:] == -1)
        if n >= 3000 and spread >= 12.0:
            assert m > max_keep and m < n       # the chain really was cut short, and boxes really were suppressed
        elif n == 6000:
            assert m < max_keep                 # ... and here it runs through every chunk without filling up
