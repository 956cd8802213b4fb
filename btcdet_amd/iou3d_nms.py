"""Rotated BEV IoU and NMS behind the reference's two interfaces:

* the functions of ``btcdet/ops/iou3d_nms/iou3d_nms_utils.py`` (:12-115) -- ``boxes_iou_bev``, ``boxes_iou3d_gpu``,
  ``nms_gpu``, ``nms_normal_gpu``, ``boxes_bev_iou_cpu`` -- same arguments and return values;
* ``iou3d_nms_cuda``: a stand-in for the reference's COMPILED module (``iou3d_nms_api.cpp``: ``boxes_overlap_bev_gpu``,
  ``boxes_iou_bev_gpu``, ``nms_gpu``, ``nms_normal_gpu``, ``boxes_iou_bev_cpu``) with the same call signatures, so the
  reference's own ``iou3d_nms_utils.py`` runs unchanged after ``install_as_iou3d_nms_cuda()``.

All arithmetic is in libbtcdet_hip.so (csrc/iou3d_nms.hip); there is no CPU path (CPU tensors are copied to the GPU)."""
import sys
import types

import torch

from ._lib import check, lib, ptr, stream_ptr, workspace


def _boxes(t, device=None):
    if t.shape[-1] != 7:
        raise ValueError("boxes must be (N, 7) [x, y, z, dx, dy, dz, heading], got %s" % (tuple(t.shape),))
    t = t.to(device=device if device is not None else t.device, dtype=torch.float32)
    if not t.is_cuda:
        t = t.cuda()
    return t.contiguous()


def _pairwise(boxes_a, boxes_b, mode, out=None):
    a, b = _boxes(boxes_a), _boxes(boxes_b)
    if out is None:
        out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    check(lib().btc_boxes_pairwise_bev(ptr(a), a.shape[0], ptr(b), b.shape[0], mode, ptr(out), stream_ptr()), "btc_boxes_pairwise_bev")
    return out


def _nms_sorted(boxes_sorted, thresh, rotated):
    """boxes sorted by descending score -> (keep positions (int64, device), number kept (python int; one 4-byte read-back))"""
    b = _boxes(boxes_sorted)
    n = b.shape[0]
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=b.device)
    cnt = torch.zeros((1,), dtype=torch.int32, device=b.device)
    ws_bytes = lib().btc_nms_ws_bytes(n)
    ws = workspace(ws_bytes, b.device)
    check(lib().btc_nms(ptr(b), n, float(thresh), int(rotated), ptr(keep), ptr(cnt), ptr(ws), ws_bytes, stream_ptr()), "btc_nms")
    return keep, int(cnt.item())


def nms_topk(boxes_sorted, thresh, max_keep, rotated=True):
    """boxes_sorted (B, n, 7) or (n, 7), every scene sorted by descending score -> (keep (B, max_keep) int64 positions, -1 padded,
    num_keep (B,) int32), both on the device: the first max_keep entries of the greedy chain, which stops there (csrc/iou3d_nms.hip
    btc_nms_topk) -- no host read-back, all scenes in shared launches"""
    b = boxes_sorted if boxes_sorted.dim() == 3 else boxes_sorted.unsqueeze(0)
    B, n = int(b.shape[0]), int(b.shape[1])
    b = _boxes(b.reshape(-1, 7)).view(B, n, 7)
    keep = torch.empty((B, int(max_keep)), dtype=torch.int64, device=b.device)
    cnt = torch.empty((B,), dtype=torch.int32, device=b.device)
    ws_bytes = lib().btc_nms_topk_ws_bytes(B, n, int(max_keep))
    ws = workspace(ws_bytes, b.device)
    check(lib().btc_nms_topk(ptr(b), B, n, float(thresh), int(bool(rotated)), int(max_keep), ptr(keep), ptr(cnt), ptr(ws), ws_bytes, stream_ptr()),
          "btc_nms_topk")
    return keep, cnt


# ---------------------------------------------------------------- iou3d_nms_utils.py surface
def boxes_iou_bev(boxes_a, boxes_b):
    """(N,7), (M,7) -> (N,M) rotated BEV IoU (iou3d_nms_utils.py:32-46)"""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    return _pairwise(boxes_a, boxes_b, 1)


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """the reference's CPU entry point (iou3d_nms_utils.py:12-29): numpy or CPU tensors in, same kind out"""
    is_numpy = not isinstance(boxes_a, torch.Tensor)
    a = torch.as_tensor(boxes_a, dtype=torch.float32)
    b = torch.as_tensor(boxes_b, dtype=torch.float32)
    out = _pairwise(a, b, 1).cpu()
    return out.numpy() if is_numpy else out


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N,7), (M,7) -> (N,M) 3-D IoU = BEV overlap x height overlap over the union volume (iou3d_nms_utils.py:48-78)"""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a, b = _boxes(boxes_a), _boxes(boxes_b)
    overlaps_bev = _pairwise(a, b, 0)
    a_hmax, a_hmin = (a[:, 2] + a[:, 5] / 2).view(-1, 1), (a[:, 2] - a[:, 5] / 2).view(-1, 1)
    b_hmax, b_hmin = (b[:, 2] + b[:, 5] / 2).view(1, -1), (b[:, 2] - b[:, 5] / 2).view(1, -1)
    overlaps_h = torch.clamp(torch.min(a_hmax, b_hmax) - torch.max(a_hmin, b_hmin), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (a[:, 3] * a[:, 4] * a[:, 5]).view(-1, 1)
    vol_b = (b[:, 3] * b[:, 4] * b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """rotated NMS (iou3d_nms_utils.py:81-96): returns (indices of the kept boxes in descending score order, None)"""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep, n = _nms_sorted(boxes[order], thresh, True)
    return order[keep[:n]].contiguous(), None


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """axis-aligned NMS on the BEV footprints (iou3d_nms_utils.py:99-115)"""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    keep, n = _nms_sorted(boxes[order], thresh, False)
    return order[keep[:n]].contiguous(), None


# ---------------------------------------------------------------- stand-in for the compiled module
def _c_boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    _pairwise(boxes_a, boxes_b, 0, ans_overlap)
    return 1


def _c_boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    _pairwise(boxes_a, boxes_b, 1, ans_iou)
    return 1


def _c_boxes_iou_bev_cpu(boxes_a, boxes_b, ans_iou):
    ans_iou.copy_(_pairwise(boxes_a, boxes_b, 1).cpu())
    return 1


def _c_nms(rotated):
    def fn(boxes, keep, nms_overlap_thresh):
        """boxes (N,7) sorted by score on the GPU, keep: int64 tensor of N entries (the reference passes a CPU tensor); returns the count"""
        k, n = _nms_sorted(boxes, nms_overlap_thresh, rotated)
        keep[:n] = k[:n].to(keep.device)
        return n
    return fn


iou3d_nms_cuda = types.ModuleType("iou3d_nms_cuda")
iou3d_nms_cuda.boxes_overlap_bev_gpu = _c_boxes_overlap_bev_gpu
iou3d_nms_cuda.boxes_iou_bev_gpu = _c_boxes_iou_bev_gpu
iou3d_nms_cuda.boxes_iou_bev_cpu = _c_boxes_iou_bev_cpu
iou3d_nms_cuda.nms_gpu = _c_nms(True)
iou3d_nms_cuda.nms_normal_gpu = _c_nms(False)


def install_as_iou3d_nms_cuda():
    """register the stand-in under the name the reference imports (iou3d_nms_utils.py:9: ``from . import iou3d_nms_cuda``)"""
    sys.modules["btcdet.ops.iou3d_nms.iou3d_nms_cuda"] = iou3d_nms_cuda
    return iou3d_nms_cuda
