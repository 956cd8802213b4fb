"""SparseMaxPool2d/3d (spconv v1.2.1): conv geometry rulebook (never cached: no indice_key) + max
reduction.  Call site: /root/reference/btcdet/models/backbones_3d/spconv_backbone.py:29,831-847."""
import torch

from . import ops
from .conv import _ntuple
from .modules import SparseModule
from .tensor import SparseConvTensor


class SparseMaxPool(SparseModule):
    def __init__(self, ndim, kernel_size, stride=1, padding=0, dilation=1, subm=False):
        super(SparseMaxPool, self).__init__()
        self.ndim = ndim
        self.kernel_size = _ntuple(kernel_size, ndim)
        self.stride = _ntuple(stride, ndim)
        self.padding = _ntuple(padding, ndim)
        self.dilation = _ntuple(dilation, ndim)
        self.subm = subm

    def _k3(self, v, fill):
        return list(v) if self.ndim == 3 else [fill] + list(v)

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        spatial_shape = [int(s) for s in input.spatial_shape]
        if not self.subm:
            out_spatial_shape = ops.get_conv_output_size(spatial_shape, self.kernel_size, self.stride, self.padding,
                                                         self.dilation)
        else:
            out_spatial_shape = spatial_shape
        indices = input.indices
        if self.ndim == 2:
            indices = torch.cat([indices[:, :1], torch.zeros_like(indices[:, :1]), indices[:, 1:]], dim=1)
            spatial_shape = [1] + spatial_shape
        rb = ops.build_rulebook(indices, input.batch_size, spatial_shape, self._k3(self.kernel_size, 1),
                                self._k3(self.stride, 1), self._k3(self.padding, 0), self._k3(self.dilation, 1), 0,
                                self.subm, False)
        out_features = ops.indice_maxpool(input.features, rb)
        outids = rb.out_indices
        if self.ndim == 2:
            outids = torch.cat([outids[:, :1], outids[:, 2:]], dim=1).contiguous()
        out_tensor = SparseConvTensor(out_features, outids, out_spatial_shape, input.batch_size)
        out_tensor.indice_dict = input.indice_dict
        out_tensor.grid = input.grid
        return out_tensor


class SparseMaxPool2d(SparseMaxPool):
    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        super(SparseMaxPool2d, self).__init__(2, kernel_size, stride, padding, dilation)


class SparseMaxPool3d(SparseMaxPool):
    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        super(SparseMaxPool3d, self).__init__(3, kernel_size, stride, padding, dilation)
