"""SparseConvTensor -- same attributes and methods the reference reads and writes
(/root/reference/btcdet/models/backbones_3d/spconv_backbone.py:155-160,207-211,872-915;
height_compression.py:21; occ_head_3D.py:46,51)."""
import numpy as np
import torch

from . import ops


class SparseConvTensor(object):
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        """features (N,C) float32, indices (N,ndim+1) int32 [b,z,y,x], spatial_shape list / ndarray."""
        self.features = features
        self.indices = indices if indices.dtype == torch.int32 else indices.int()
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        self.indice_dict = {}  # shared BY REFERENCE with every tensor derived from this one (App. B.5)
        self.grid = grid

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key, None)

    def dense(self, channels_first=True):
        if self.features.dtype == torch.bfloat16:  # dense maps (head logits, BEV map) are fp32; the widening is exact
            t = SparseConvTensor(self.features.float(), self.indices, self.spatial_shape, self.batch_size)
            return t.dense(channels_first)
        idx = self.indices
        shape = [int(v) for v in self.spatial_shape]
        if idx.shape[1] == 3:  # 2-D tensor [b,y,x]: densify as depth-1 volume then drop the axis
            idx4 = torch.cat([idx[:, :1], torch.zeros_like(idx[:, :1]), idx[:, 1:]], dim=1)
            res = ops.ToDenseFunction.apply(self.features, idx4, self.batch_size, [1] + shape).squeeze(2)
        else:
            res = ops.ToDenseFunction.apply(self.features, idx, self.batch_size, shape)
        if channels_first:
            return res
        ndim = len(shape)
        return res.permute(0, *range(2, ndim + 2), 1).contiguous()

    @property
    def sparity(self):
        return self.indices.shape[0] / float(max(self.spatial_size, 1)) / float(max(self.batch_size, 1))
