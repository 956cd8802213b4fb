"""A CPU stand-in for `spconv` v1.2.1 backed by the C oracle (oracle/btc_oracle.c), so that the REFERENCE's own model files
(spconv_backbone.py, occ_head_3D.py, height_compression.py) execute in this container and emit golden vectors for the parts
of the hot path whose graph lives in the reference but whose primitive lives in the absent third-party library.
What this pins: the reference's layer graph / indice_key reuse / tensor plumbing, executed by the reference's code.  What it
cannot pin: the sparse-conv primitive itself (that is the oracle, "parity unpinned").
Forward only (torch.no_grad); float32 features through oracle.conv_fwd / maxpool_fwd / dense.  Generator-side test
infrastructure: imported by tests/golden/gen_golden_full.py and nothing else.  Semantics follow SURVEY.md App. B."""
import math
import types

import numpy as np
import torch
from torch import nn

from oracle import oracle as orc


def _t3(v, n=3):
    if isinstance(v, (list, tuple, np.ndarray)):
        return [int(x) for x in v]
    return [int(v)] * n


class SparseConvTensor(object):
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        self.indice_dict = {}
        self.grid = grid

    def find_indice_pair(self, key):
        return None if key is None else self.indice_dict.get(key, None)

    def dense(self, channels_first=True):
        shape = [int(v) for v in self.spatial_shape]
        idx = self.indices.numpy().astype(np.int32)
        if idx.shape[1] == 3:
            idx = np.concatenate([idx[:, :1], np.zeros_like(idx[:, :1]), idx[:, 1:]], axis=1)
            out = orc.dense(self.features.detach().numpy(), idx, int(self.batch_size), [1] + shape)[:, :, 0]
        else:
            out = orc.dense(self.features.detach().numpy(), idx, int(self.batch_size), shape)
        out = torch.from_numpy(out)
        if channels_first:
            return out
        return out.permute(0, *range(2, out.dim()), 1).contiguous()


class SparseModule(nn.Module):
    pass


class SparseSequential(SparseModule):
    def __init__(self, *args, **kwargs):
        super().__init__()
        for i, m in enumerate(args):
            self.add_module(str(i), m)
        for k, m in kwargs.items():
            self.add_module(k, m)

    def __getitem__(self, i):
        return list(self._modules.values())[i]

    def __len__(self):
        return len(self._modules)

    def forward(self, input):
        for m in self._modules.values():
            if isinstance(m, SparseModule):
                input = m(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    input.features = m(input.features)
            else:
                input = m(input)
        return input


class _Conv(SparseModule):
    def __init__(self, ndim, cin, cout, kernel_size=3, stride=1, padding=0, dilation=1, groups=1, bias=True, subm=False,
                 output_padding=0, transposed=False, inverse=False, indice_key=None):
        super().__init__()
        assert groups == 1 and ndim == 3
        self.k, self.s, self.p, self.d, self.op = _t3(kernel_size), _t3(stride), _t3(padding), _t3(dilation), _t3(output_padding)
        self.subm, self.transposed, self.inverse, self.indice_key = subm, transposed, inverse, indice_key
        self.weight = nn.Parameter(torch.empty(*self.k, cin, cout))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            nn.init.zeros_(self.bias)

    def forward(self, x):
        rb = x.find_indice_pair(self.indice_key)
        idx = x.indices.numpy().astype(np.int32)
        shape = [int(v) for v in x.spatial_shape]
        if self.inverse:
            assert rb is not None
            in_idx, in_shape, o_idx, nbr_out, nbr_in, osh = rb
            maps, outids, out_shape = nbr_in, in_idx, in_shape
        else:
            if rb is None:
                mode = orc.MODE_SUBM if self.subm else (orc.MODE_TRANSPOSE if self.transposed else orc.MODE_CONV)
                o_idx, nbr_out, nbr_in, osh = orc.rulebook(idx, shape, self.k, self.s, self.p, self.d, mode, self.op)
                rb = (idx, shape, o_idx, nbr_out, nbr_in, [int(v) for v in osh])
                if self.indice_key is not None:
                    x.indice_dict[self.indice_key] = rb
            maps, outids, out_shape = rb[3], rb[2], rb[5]    # cache hit: used without checking this layer's geometry (App. B.5)
        b = None if self.bias is None else self.bias.detach().numpy()
        f = orc.conv_fwd(x.features.detach().numpy(), self.weight.detach().numpy(), b, maps)
        out = SparseConvTensor(torch.from_numpy(f), torch.from_numpy(np.ascontiguousarray(outids)), out_shape, x.batch_size)
        out.indice_dict = x.indice_dict
        out.grid = x.grid
        return out


class SubMConv3d(_Conv):
    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True, indice_key=None):
        super().__init__(3, cin, cout, kernel_size, stride, padding, dilation, groups, bias, True, indice_key=indice_key)


class SparseConv3d(_Conv):
    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True, indice_key=None):
        super().__init__(3, cin, cout, kernel_size, stride, padding, dilation, groups, bias, indice_key=indice_key)


class SparseConvTranspose3d(_Conv):
    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True, indice_key=None):
        super().__init__(3, cin, cout, kernel_size, stride, padding, dilation, groups, bias, transposed=True, indice_key=indice_key)


class SparseInverseConv3d(_Conv):
    def __init__(self, cin, cout, kernel_size, indice_key, bias=True):
        super().__init__(3, cin, cout, kernel_size, bias=bias, inverse=True, indice_key=indice_key)


class SubMConv2d(SparseModule):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("SubMConv2d is not on the configured path")


class SparseMaxPool3d(SparseModule):
    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__()
        self.k, self.s, self.p, self.d = _t3(kernel_size), _t3(stride), _t3(padding), _t3(dilation)

    def forward(self, x):
        idx = x.indices.numpy().astype(np.int32)
        o_idx, nbr_out, nbr_in, osh = orc.rulebook(idx, [int(v) for v in x.spatial_shape], self.k, self.s, self.p, self.d, orc.MODE_CONV)
        f = orc.maxpool_fwd(x.features.detach().numpy(), nbr_out)
        out = SparseConvTensor(torch.from_numpy(f), torch.from_numpy(o_idx), [int(v) for v in osh], x.batch_size)
        out.indice_dict = x.indice_dict
        out.grid = x.grid
        return out


utils = types.ModuleType("spconv.utils")
utils.VoxelGeneratorV2 = orc.VoxelGeneratorV2
utils.VoxelGenerator = orc.VoxelGeneratorV2
