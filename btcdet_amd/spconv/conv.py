"""Sparse convolution layers with spconv v1.2.1's constructor signatures, parameter names and
shapes (weight [kD,kH,kW,Cin,Cout], optional bias [Cout]) so reference checkpoints load
(/root/reference/btcdet/models/detectors/detector3d_template.py:594-618) and the reference's
model code constructs them unchanged (spconv_backbone.py:12-29,58-64,657,696; occ_head_3D.py:26,31)."""
import math

import numpy as np
import torch
from torch import nn
from torch.nn import init

from . import ops
from .modules import SparseModule
from .tensor import SparseConvTensor


def _ntuple(v, n):
    if isinstance(v, (list, tuple, np.ndarray)):
        v = [int(x) for x in v]
        assert len(v) == n, f"expected {n} values, got {v}"
        return v
    return [int(v)] * n


def geometry_key(indices, spatial_shape, kernel_size, dilation, subm, transposed, stride, padding, output_padding):
    key = (indices.data_ptr(), tuple(indices.shape), tuple(int(v) for v in spatial_shape), tuple(int(v) for v in kernel_size),
           tuple(int(v) for v in dilation), bool(subm), bool(transposed))
    if not subm:  # stride / padding do not enter a submanifold rulebook
        key = key + (tuple(int(v) for v in stride), tuple(int(v) for v in padding), tuple(int(v) for v in output_padding))
    return key


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None,
                 fused_bn=False):
        super(SparseConvolution, self).__init__()
        assert groups == 1, "groups != 1 is not supported (nor used by the reference)"
        assert ndim in (2, 3)
        self.ndim = ndim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _ntuple(kernel_size, ndim)
        self.conv1x1 = int(np.prod(self.kernel_size)) == 1
        self.stride = _ntuple(stride, ndim)
        self.padding = _ntuple(padding, ndim)
        self.dilation = _ntuple(dilation, ndim)
        self.output_padding = _ntuple(output_padding, ndim)
        self.transposed, self.inverse, self.subm = transposed, inverse, subm
        self.groups = groups
        self.indice_key = indice_key
        self.fused_bn = fused_bn
        self.weight = nn.Parameter(torch.Tensor(*self.kernel_size, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        # same calls as spconv v1.2.1 (including its fan-in quirk on the [k..,Cin,Cout] layout)
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def forward(self, input, fuse_bn=None):
        """fuse_bn = (BatchNorm1d, relu: bool): SparseSequential hands over the BatchNorm (+ ReLU) that follows this layer so
        that conv -> BN -> ReLU is one autograd node (ops.indice_conv_bn_relu); the modules and their parameters are untouched"""
        assert isinstance(input, SparseConvTensor)
        features, indices = input.features, input.indices
        spatial_shape, batch_size = input.spatial_shape, input.batch_size
        out_spatial_shape = self._out_shape(spatial_shape)
        if self.conv1x1:
            features = torch.mm(input.features, self.weight.view(self.in_channels, self.out_channels))
            if self.bias is not None:
                features = features + self.bias
            if fuse_bn is not None and features.shape[0] > 0:
                from . import fused_bn
                features = fused_bn.batch_norm_relu(fuse_bn[0], features, fuse_bn[1])
            out_tensor = SparseConvTensor(features, input.indices, input.spatial_shape, input.batch_size)
            out_tensor.indice_dict = input.indice_dict
            out_tensor.grid = input.grid
            return out_tensor
        rb, outids, out_spatial_shape = self._resolve_rulebook(input, out_spatial_shape)
        out_features = self._conv_apply(features, rb, self.inverse, fuse_bn)
        out_tensor = SparseConvTensor(out_features, outids, out_spatial_shape, batch_size)
        out_tensor.indice_dict = input.indice_dict
        out_tensor.grid = input.grid
        return out_tensor

    def _resolve_rulebook(self, input, out_spatial_shape):
        """the geometry half of forward: this layer's rulebook (cached under indice_key / the geometry cache, or built now), the
        output index tensor and spatial shape; announces the next strided layers (lookahead)"""
        indices, spatial_shape, batch_size = input.indices, input.spatial_shape, input.batch_size
        rb = input.find_indice_pair(self.indice_key)
        if self.inverse:
            assert rb is not None and self.indice_key is not None, "inverse conv needs the cached rulebook of its indice_key"
            assert rb.K == int(np.prod(self.kernel_size)), "inverse conv kernel does not match the cached rulebook"
            return rb, rb.in_indices, rb.in_shape[3 - self.ndim:]
        if rb is None:
            rb = self._rulebook(indices, spatial_shape, batch_size, input.indice_dict)
            if self.indice_key is not None:
                input.indice_dict[self.indice_key] = rb
        # on a cache hit the layer uses the cached rulebook without checking its own geometry (App. B.5)
        outids = rb.out_indices
        if self.ndim == 2 and outids.shape[1] == 4:
            outids = torch.cat([outids[:, :1], outids[:, 2:]], dim=1).contiguous()
        # the strided layers that consume this output level start counting their rows now, on the side stream,
        # while this layer's (and the following submanifold layers') feature kernels run (ops.py, LOOKAHEAD)
        for nxt in getattr(self, "lookahead", ()):
            nxt.prefetch(outids, out_spatial_shape, batch_size, input.indice_dict)
        return rb, outids, out_spatial_shape

    def forward_geometry(self, input):
        """forward without the feature kernels: builds (and caches in input.indice_dict) what forward would build and returns a
        feature-less SparseConvTensor on the output active set.  A later forward over the same indice_dict finds every
        rulebook ready (BtcHotPath.prepare: the occupancy branch's geometry is a function of the input coordinates only)"""
        if self.conv1x1:
            return input
        rb, outids, out_spatial_shape = self._resolve_rulebook(input, self._out_shape(input.spatial_shape))
        out_tensor = SparseConvTensor(None, outids, out_spatial_shape, input.batch_size)
        out_tensor.indice_dict = input.indice_dict
        out_tensor.grid = input.grid
        return out_tensor

    def _out_shape(self, spatial_shape):
        """output spatial shape for an input shape (memoised per layer: the grids of a model are fixed)"""
        key = tuple(int(v) for v in spatial_shape)
        memo = self.__dict__.setdefault("_out_shape_memo", {})
        hit = memo.get(key)
        if hit is None:
            if self.subm:
                hit = spatial_shape
            elif self.transposed:
                hit = ops.get_deconv_output_size(spatial_shape, self.kernel_size, self.stride, self.padding, self.dilation, self.output_padding)
            else:
                hit = ops.get_conv_output_size(spatial_shape, self.kernel_size, self.stride, self.padding, self.dilation)
            memo[key] = hit
        return hit

    def _conv_apply(self, features, rb, inverse, fuse_bn):
        n_res = rb.n_in if inverse else rb.n_out
        if fuse_bn is not None and n_res > 0:
            return ops.indice_conv_bn_relu(features, self.weight, self.bias, rb, fuse_bn[0], fuse_bn[1], inverse=inverse)
        return ops.indice_conv(features, self.weight, self.bias, rb, inverse=inverse)  # empty result: BatchNorm is skipped, as SparseSequential does

    # A rulebook is a pure function of (indices, geometry): layers that ask for the same geometry on the same index
    # tensor under different indice_keys (OccHead3D's 'cls_ind' / 'res_ind' after 'subm5', occ_head_3D.py:26,31; a
    # SparseMaxPool3d beside a SparseConv3d, spconv_backbone.py:831-847) share one build.  The cache lives in the shared
    # indice_dict and keeps the index tensor alive, so a recycled data_ptr can never alias.
    def _gkey(self, indices, spatial_shape):
        return geometry_key(indices, spatial_shape, self.kernel_size, self.dilation, self.subm, self.transposed, self.stride,
                            self.padding, self.output_padding)

    def _rulebook(self, indices, spatial_shape, batch_size, indice_dict):
        geom = indice_dict.setdefault("__geometry_cache__", {})
        gkey = self._gkey(indices, spatial_shape)
        hit = geom.get(gkey, None)
        if hit is not None:
            rb = hit[0]
            if isinstance(rb, ops.PendingRulebook):
                rb = rb.finish()
                geom[gkey] = (rb, indices)
            return rb
        idx4 = indices
        if self.ndim == 2:
            idx4 = torch.cat([indices[:, :1], torch.zeros_like(indices[:, :1]), indices[:, 1:]], dim=1)
        rb = ops.build_rulebook_g(idx4, batch_size, self._geometry(spatial_shape))
        geom[gkey] = (rb, indices)
        return rb

    def _geometry(self, spatial_shape):
        """ops._Geometry of this layer for an input shape (memoised: host-side constants and ctypes pointers)"""
        key = tuple(int(v) for v in spatial_shape)
        memo = self.__dict__.setdefault("_geometry_memo", {})
        g = memo.get(key)
        if g is None:
            g = memo[key] = ops._geometry(self._shape3(spatial_shape), self._k3(self.kernel_size, 1), self._k3(self.stride, 1),
                                          self._k3(self.padding, 0), self._k3(self.dilation, 1), self._k3(self.output_padding, 0), self.subm,
                                          self.transposed)
        return g

    def prefetch(self, indices, spatial_shape, batch_size, indice_dict):
        """start the count half of this layer's rulebook for `indices` (no-op for submanifold / inverse / 2-D / cached layers)"""
        if self.subm or self.inverse or self.conv1x1 or self.ndim != 3 or not indices.is_cuda:
            return
        if not ops.lookahead_enabled() or (self.indice_key is not None and self.indice_key in indice_dict):
            return
        geom = indice_dict.setdefault("__geometry_cache__", {})
        gkey = self._gkey(indices, spatial_shape)
        if gkey in geom:
            return
        geom[gkey] = (ops.prefetch_conv_rulebook(indices, batch_size, spatial_shape, self.kernel_size, self.stride, self.padding,
                                                 self.dilation, self.output_padding, self.transposed), indices)

    def _k3(self, v, fill):
        return list(v) if self.ndim == 3 else [fill] + list(v)

    def _shape3(self, shape):
        shape = [int(s) for s in shape]
        return shape if self.ndim == 3 else [1] + shape


class SparseConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super(SparseConv2d, self).__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                                           bias, indice_key=indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super(SparseConv3d, self).__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                                           bias, indice_key=indice_key)


class SparseConvTranspose2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super(SparseConvTranspose2d, self).__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation,
                                                    groups, bias, transposed=True, indice_key=indice_key)


class SparseConvTranspose3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super(SparseConvTranspose3d, self).__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                                                    groups, bias, transposed=True, indice_key=indice_key)


class SparseInverseConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
        super(SparseInverseConv2d, self).__init__(2, in_channels, out_channels, kernel_size, bias=bias, inverse=True,
                                                  indice_key=indice_key)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
        super(SparseInverseConv3d, self).__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True,
                                                  indice_key=indice_key)


class SubMConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super(SubMConv2d, self).__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                                         True, indice_key=indice_key)


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super(SubMConv3d, self).__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                                         True, indice_key=indice_key)
