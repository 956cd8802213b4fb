"""pointnet2_stack on MI355X: the operator surface of
/root/reference/btcdet/ops/pointnet2/pointnet2_stack/pointnet2_utils.py (ball_query, grouping_operation, QueryAndGroup,
furthest_point_sample, three_nn, three_interpolate) and pointnet2_modules.py (StackSAModuleMSG, StackPointnetFPModule) over the
HIP kernels of csrc/pointnet2.hip -- same names, argument order, return values and quirks -- plus `pointnet2_stack_cuda`, a
stand-in for the reference's compiled module with its eight `*_wrapper` entry points (src/pointnet2_api.cpp), so that the
reference's own pointnet2_utils.py binds it unchanged.  Consumers in the reference: the ROI head's grid
pooling (models/roi_heads/conv_head.py:53-70,283-343) and the point feature encoders (backbones_3d/pfe)."""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from ._lib import check, lib, ptr, stream_ptr


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("pointnet2_stack: the HIP kernels need CUDA (ROCm) tensors; there is no CPU fallback")


class _Cuda(object):
    """`pointnet2_stack_cuda` of the reference: same function names and positional arguments, results written in place"""

    @staticmethod
    def ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx):
        _need_cuda(new_xyz, xyz, idx)
        check(lib().btc_ball_query(ptr(new_xyz), ptr(new_xyz_batch_cnt), ptr(xyz), ptr(xyz_batch_cnt), int(B), int(M), -1.0, float(radius),
                                   int(nsample), ptr(idx), stream_ptr()), "btc_ball_query")

    @staticmethod
    def shell_query_wrapper(B, M, inner_radius, outer_radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx):
        _need_cuda(new_xyz, xyz, idx)
        check(lib().btc_ball_query(ptr(new_xyz), ptr(new_xyz_batch_cnt), ptr(xyz), ptr(xyz_batch_cnt), int(B), int(M), float(inner_radius),
                                   float(outer_radius), int(nsample), ptr(idx), stream_ptr()), "btc_ball_query")

    @staticmethod
    def group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out):
        _need_cuda(features, idx, out)
        check(lib().btc_group_points(ptr(features), ptr(features_batch_cnt), ptr(idx), ptr(idx_batch_cnt), int(B), int(M), int(C), int(nsample),
                                     ptr(out), stream_ptr()), "btc_group_points")

    @staticmethod
    def group_points_grad_wrapper(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features):
        _need_cuda(grad_out, idx, grad_features)
        check(lib().btc_group_points_grad(ptr(grad_out), ptr(idx), ptr(idx_batch_cnt), ptr(features_batch_cnt), int(B), int(M), int(C), int(N),
                                          int(nsample), ptr(grad_features), stream_ptr()), "btc_group_points_grad")

    @staticmethod
    def furthest_point_sampling_wrapper(B, N, npoint, xyz, temp, out):
        _need_cuda(xyz, temp, out)
        check(lib().btc_furthest_point_sampling(ptr(xyz), int(B), int(N), int(npoint), ptr(temp), ptr(out), stream_ptr()),
              "btc_furthest_point_sampling")

    @staticmethod
    def three_nn_wrapper(unknown, unknown_batch_cnt, known, known_batch_cnt, dist2, idx):
        _need_cuda(unknown, known, dist2, idx)
        check(lib().btc_three_nn(ptr(unknown), ptr(unknown_batch_cnt), ptr(known), ptr(known_batch_cnt), int(unknown_batch_cnt.shape[0]),
                                 int(unknown.shape[0]), ptr(dist2), ptr(idx), stream_ptr()), "btc_three_nn")

    @staticmethod
    def three_interpolate_wrapper(features, idx, weight, out):
        _need_cuda(features, idx, weight, out)
        check(lib().btc_three_interpolate(ptr(features), ptr(idx), ptr(weight), int(idx.shape[0]), int(features.shape[1]), ptr(out),
                                          stream_ptr()), "btc_three_interpolate")

    @staticmethod
    def three_interpolate_grad_wrapper(grad_out, idx, weight, grad_features):
        _need_cuda(grad_out, idx, weight, grad_features)
        check(lib().btc_three_interpolate_grad(ptr(grad_out), ptr(idx), ptr(weight), int(grad_out.shape[0]), int(grad_out.shape[1]),
                                               int(grad_features.shape[0]), ptr(grad_features), stream_ptr()), "btc_three_interpolate_grad")


pointnet2_stack_cuda = _Cuda()
pointnet2 = pointnet2_stack_cuda


def install_as_pointnet2_stack_cuda(package="btcdet.ops.pointnet2.pointnet2_stack"):
    """register the stand-in under the name the reference imports (`from . import pointnet2_stack_cuda`)"""
    import sys
    import types
    mod = types.ModuleType(package + ".pointnet2_stack_cuda")
    for name in dir(_Cuda):
        if name.endswith("_wrapper"):
            setattr(mod, name, getattr(_Cuda, name))
    sys.modules[package + ".pointnet2_stack_cuda"] = mod
    return mod


def _contig(*tensors):
    for t in tensors:
        if not t.is_contiguous():
            raise AssertionError("pointnet2_stack: contiguous tensors expected")


class BallQuery(Function):
    """ball (radius: float) or shell (radius: [inner, outer]) query over stacked scenes -> (idx, empty_ball_mask);
    idx (M, nsample) int32 is LOCAL to each query's scene, balls without a hit are zeroed and flagged (pointnet2_utils.py:11-40)"""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
        _contig(new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt)
        n_scene, n_query = xyz_batch_cnt.shape[0], new_xyz.shape[0]
        idx = torch.empty((n_query, nsample), dtype=torch.int32, device=new_xyz.device)  # the C ABI call zero-fills it
        if isinstance(radius, (list, tuple)):
            pointnet2.shell_query_wrapper(n_scene, n_query, radius[0], radius[1], nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx)
        else:
            pointnet2.ball_query_wrapper(n_scene, n_query, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx)
        empty = idx[:, 0] == -1
        idx[empty] = 0
        ctx.mark_non_differentiable(idx, empty)
        return idx, empty

    @staticmethod
    def backward(ctx, *grads):
        return (None,) * 6


ball_query = BallQuery.apply


def host_counts(cnt):
    """the per-scene counts of a *_batch_cnt tensor as python ints.  The reference checks them against the row counts (and branches on
    them) at every grouping call -- 50-odd blocking read-backs per training step of the ROI head when the tensors live on the GPU.
    Here a count tensor carries its host copy (`_btc_host`, attached by whoever made it or by the first call) so that ONE read-back
    serves every check of that tensor."""
    h = getattr(cnt, "_btc_host", None)
    if h is None:
        h = [int(v) for v in cnt.tolist()]
        try:
            cnt._btc_host = h
        except AttributeError:
            pass
    return h


def with_host_counts(cnt, host):
    """attach host-known counts to a count tensor (no read-back will be needed for it)"""
    cnt._btc_host = [int(v) for v in host]
    return cnt


class GroupingOperation(Function):
    """features (N1+N2.., C) gathered at idx (M1+M2.., nsample) -> (M1+M2.., C, nsample); the backward scatter-adds
    (pointnet2_utils.py:52-104)"""

    @staticmethod
    def forward(ctx, features, features_batch_cnt, idx, idx_batch_cnt):
        _contig(features, features_batch_cnt, idx, idx_batch_cnt)
        if features.shape[0] != sum(host_counts(features_batch_cnt)):
            raise AssertionError('features: %s, features_batch_cnt: %s' % (str(features.shape), str(features_batch_cnt)))
        if idx.shape[0] != sum(host_counts(idx_batch_cnt)):
            raise AssertionError('idx: %s, idx_batch_cnt: %s' % (str(idx.shape), str(idx_batch_cnt)))
        (n_query, nsample), (n_pts, ch) = idx.shape, features.shape
        n_scene = idx_batch_cnt.shape[0]
        grouped = torch.empty((n_query, ch, nsample), dtype=torch.float32, device=features.device)
        pointnet2.group_points_wrapper(n_scene, n_query, ch, nsample, features, features_batch_cnt, idx, idx_batch_cnt, grouped)
        ctx.for_backwards = (n_scene, n_pts, idx, features_batch_cnt, idx_batch_cnt)
        return grouped

    @staticmethod
    def backward(ctx, grad_out):
        n_scene, n_pts, idx, features_batch_cnt, idx_batch_cnt = ctx.for_backwards
        n_query, ch, nsample = grad_out.shape
        grad_features = torch.empty((n_pts, ch), dtype=torch.float32, device=grad_out.device)  # zero-filled by the call
        pointnet2.group_points_grad_wrapper(n_scene, n_query, ch, n_pts, nsample, grad_out.contiguous(), idx, idx_batch_cnt, features_batch_cnt,
                                            grad_features)
        return grad_features, None, None, None


grouping_operation = GroupingOperation.apply


class QueryAndGroup(nn.Module):
    """ball query + grouping of the offsets (and features) around every query, with the reference's extras
    (pointnet2_utils.py:111-188): rotation of the offsets by a per-roi matrix, division by per-roi xy / z scales, zeroed
    empty balls, and its handling of a trailing scene without points (whose queries get zero groups)"""

    def __init__(self, radius, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    @staticmethod
    def _group(values, cnt, idx, qcnt, drop_last):
        if not drop_last:
            return grouping_operation(values, cnt, idx, qcnt)
        n_last = host_counts(qcnt)[-1]
        g = grouping_operation(values, with_host_counts(cnt[0:-1], host_counts(cnt)[0:-1]), idx[:-n_last],
                               with_host_counts(qcnt[0:-1], host_counts(qcnt)[0:-1]))
        return torch.cat([g, torch.zeros_like(g[:n_last])], dim=0)

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None, rotateMatrix=None, xyscales=None, zscales=None):
        h_xyz, h_new = host_counts(xyz_batch_cnt), host_counts(new_xyz_batch_cnt)
        assert xyz.shape[0] == sum(h_xyz), 'xyz: %s, xyz_batch_cnt: %s' % (str(xyz.shape), str(new_xyz_batch_cnt))
        assert new_xyz.shape[0] == sum(h_new), 'new_xyz: %s, new_xyz_batch_cnt: %s' % (str(new_xyz.shape), str(new_xyz_batch_cnt))
        idx, empty = ball_query(self.radius, self.nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
        several = len(xyz_batch_cnt) > 1
        offsets = self._group(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt, several and h_xyz[-1] == 0)  # (M, 3, nsample)
        offsets = offsets - new_xyz.unsqueeze(-1)
        offsets[empty] = 0
        unrotated = offsets
        if rotateMatrix is not None:
            offsets = self.rotate(offsets, rotateMatrix)
        if xyscales is not None:
            offsets = torch.cat([offsets[..., :2, :] / xyscales, offsets[..., 2:3, :] / zscales], dim=-2)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            out = offsets
        else:
            # (sic) the reference tests scene 1 here and the last scene above
            grouped = self._group(features, xyz_batch_cnt, idx, new_xyz_batch_cnt, several and h_xyz[1] == 0)
            grouped[empty] = 0
            out = torch.cat([offsets, grouped], dim=1) if self.use_xyz else grouped
        return (out, idx, unrotated) if rotateMatrix is not None else (out, idx)

    def rotate(self, grouped_xyz, rotateMatrix):
        n_roi, n_query = rotateMatrix.shape[0], grouped_xyz.shape[0]
        per_query = rotateMatrix.view(n_roi, 1, 3, 3).repeat(1, n_query // n_roi, 1, 1).view(n_query, 3, 3)
        return torch.einsum("nmj,nij->nmi", grouped_xyz.permute(0, 2, 1), per_query).permute(0, 2, 1)


class FurthestPointSampling(Function):
    """xyz (B, N, 3) -> indices (B, npoint) int32 of an iterative farthest-point subset that starts at point 0
    (pointnet2_utils.py:194-213)"""

    @staticmethod
    def forward(ctx, xyz, npoint):
        _contig(xyz)
        n_scene, n_pts = xyz.shape[0], xyz.shape[1]
        picked = torch.empty((n_scene, npoint), dtype=torch.int32, device=xyz.device)
        running = torch.full((n_scene, n_pts), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2.furthest_point_sampling_wrapper(n_scene, n_pts, npoint, xyz, running, picked)
        ctx.mark_non_differentiable(picked)
        return picked

    @staticmethod
    def backward(ctx, *grads):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class ThreeNN(Function):
    """for every `unknown` point the three nearest `known` points of its scene -> (distances (N,3), global row indices (N,3))
    (pointnet2_utils.py:222-247)"""

    @staticmethod
    def forward(ctx, unknown, unknown_batch_cnt, known, known_batch_cnt):
        assert unknown.dim() == 2 and unknown.shape[1] == 3 and known.dim() == 2 and known.shape[1] == 3
        assert len(unknown_batch_cnt) == len(known_batch_cnt)
        dist2 = unknown.new_zeros(unknown.shape)
        idx = unknown_batch_cnt.new_zeros(unknown.shape).int()
        pointnet2.three_nn_wrapper(unknown.contiguous(), unknown_batch_cnt.contiguous(), known.contiguous(), known_batch_cnt.contiguous(), dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, *grads):
        return (None,) * 4


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """out[n] = sum_j weight[n, j] * features[idx[n, j]] -> (N1+N2.., C); the backward scatter-adds (pointnet2_utils.py:256-289)"""

    @staticmethod
    def forward(ctx, features, idx, weight):
        assert idx.shape[0] == weight.shape[0] and idx.shape[1] == weight.shape[1] == 3
        ctx.three_interpolate_for_backward = (idx, weight, features.shape[0])
        out = features.new_empty((idx.shape[0], features.shape[1]))
        pointnet2.three_interpolate_wrapper(features.contiguous(), idx.contiguous(), weight.contiguous(), out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, n_known = ctx.three_interpolate_for_backward
        grad_features = grad_out.new_empty((n_known, grad_out.shape[1]))  # zero-filled by the call
        pointnet2.three_interpolate_grad_wrapper(grad_out.contiguous(), idx.contiguous(), weight.contiguous(), grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


def _shared_mlp(widths):
    """1x1 Conv2d (no bias) + BatchNorm2d + ReLU per consecutive pair of widths -- the per-point MLP both modules apply to a
    (1, C, rows, samples) image; conv weights Kaiming-normal, BatchNorm at identity (what pointnet2_modules.py:43-51 sets)"""
    layers = []
    for cin, cout in zip(widths[:-1], widths[1:]):
        conv, bn = nn.Conv2d(cin, cout, kernel_size=1, bias=False), nn.BatchNorm2d(cout)
        nn.init.kaiming_normal_(conv.weight)
        nn.init.ones_(bn.weight)
        nn.init.zeros_(bn.bias)
        layers += [conv, bn, nn.ReLU()]
    return nn.Sequential(*layers)


_POOLS = {"max_pool": lambda t: t.amax(dim=3), "avg_pool": lambda t: t.mean(dim=3)}


class StackSAModuleMSG(nn.Module):
    """Set abstraction with multi-scale grouping on stacked batches: for every (radius, nsample, mlp) scale, ball-query the
    neighbours of each new point, run the shared MLP over the grouped (offset xyz + features) columns and pool over the
    samples; the scales' outputs are concatenated.  Constructor keywords, submodule names (`groupers`, `mlps`) and the forward
    signature are those the ROI head uses (/root/reference/btcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py:10-111;
    conv_head.py:117-126), so its checkpoints load."""

    def __init__(self, *, radii: List[float], nsamples: List[int], mlps: List[List[int]], use_xyz: bool = True, pool_method='max_pool'):
        super().__init__()
        if not (len(radii) == len(nsamples) == len(mlps)):
            raise ValueError("one radius, sample count and MLP spec per scale")
        if pool_method not in _POOLS:
            raise NotImplementedError(pool_method)
        self.pool_method = pool_method
        self.groupers, self.mlps = nn.ModuleList(), nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            if use_xyz:
                spec[0] += 3    # the caller's list is widened in place by the xyz offsets -- callers of the reference rely on that
            self.groupers.append(QueryAndGroup(radius, nsample, use_xyz=use_xyz))
            self.mlps.append(_shared_mlp(spec))

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None, empty_voxel_set_zeros=True, rotateMatrix=None, xyscales=None,
                zscales=None, vis=False):
        pool = _POOLS[self.pool_method]
        per_scale, seen, seen_prerot = [], [], []
        first_scene = new_xyz_batch_cnt.tolist() if vis else None
        for grouper, mlp in zip(self.groupers, self.mlps):
            res = grouper(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, rotateMatrix=rotateMatrix, xyscales=xyscales, zscales=zscales)
            grouped = res[0]                                              # (rows, C, nsample)
            # packed: MIOpen has no tuned solver for a strided view and falls back to its naive kernels (22 / 17 / 10 ms per weight-
            # gradient / data-gradient / forward call of these 1 x 1 convolutions at the ROI head's 6912 x 64 samples)
            image = grouped.permute(1, 0, 2).unsqueeze(0).contiguous()    # (1, C, rows, nsample)
            if vis:
                seen.append(torch.split(image[0].permute(1, 2, 0)[..., :3], first_scene)[0])
                if rotateMatrix is not None:
                    seen_prerot.append(torch.split(res[2].permute(0, 2, 1), first_scene)[0])
            per_scale.append(pool(mlp(image))[0].t())                     # (rows, C_out)
        out = torch.cat(per_scale, dim=1)
        return (new_xyz, out, [seen, seen_prerot]) if vis else (new_xyz, out)


class StackPointnetFPModule(nn.Module):
    """Feature propagation: every `unknown` point takes the inverse-distance weighted mean of the features of its three
    nearest `known` points (same scene), optionally concatenated with its own features, through a shared MLP
    (pointnet2_modules.py:114-153)."""

    def __init__(self, *, mlp: List[int]):
        super().__init__()
        self.mlp = _shared_mlp(mlp)

    def forward(self, unknown, unknown_batch_cnt, known, known_batch_cnt, unknown_feats=None, known_feats=None):
        dist, idx = three_nn(unknown, unknown_batch_cnt, known, known_batch_cnt)
        w = (dist + 1e-8).reciprocal()
        w = w / w.sum(dim=-1, keepdim=True)
        feats = three_interpolate(known_feats, idx, w)
        if unknown_feats is not None:
            feats = torch.cat([feats, unknown_feats], dim=1)
        return self.mlp(feats.t()[None, :, :, None])[0, :, :, 0].t()
