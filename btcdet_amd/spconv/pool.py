"""SparseMaxPool2d/3d (spconv v1.2.1): conv geometry rulebook (never cached: no indice_key) + max
reduction.  Call site: /root/reference/btcdet/models/backbones_3d/spconv_backbone.py:29,831-847."""
import torch

from . import ops
from .conv import _ntuple, geometry_key
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
        # same geometry cache as the convolutions (conv.py): a pool beside a SparseConv3d with the same kernel / stride /
        # padding on the same index tensor shares its rulebook (and a prefetched count half is finished here)
        geom = input.indice_dict.setdefault("__geometry_cache__", {})
        gkey = geometry_key(indices, spatial_shape, self.kernel_size, self.dilation, self.subm, False, self.stride, self.padding,
                            [0] * self.ndim)
        hit = geom.get(gkey, None)
        if hit is not None:
            rb = hit[0]
            if isinstance(rb, ops.PendingRulebook):
                rb = rb.finish()
                geom[gkey] = (rb, indices)
        else:
            idx4, shape3 = indices, spatial_shape
            if self.ndim == 2:
                idx4 = torch.cat([indices[:, :1], torch.zeros_like(indices[:, :1]), indices[:, 1:]], dim=1)
                shape3 = [1] + spatial_shape
            rb = ops.build_rulebook(idx4, input.batch_size, shape3, self._k3(self.kernel_size, 1), self._k3(self.stride, 1),
                                    self._k3(self.padding, 0), self._k3(self.dilation, 1), 0, self.subm, False)
            geom[gkey] = (rb, indices)
        out_features = ops.indice_maxpool(input.features, rb)
        outids = rb.out_indices
        if self.ndim == 2 and outids.shape[1] == 4:
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
