"""SparseModule / SparseSequential (spconv v1.2.1 semantics):
sparse modules receive the SparseConvTensor; plain nn.Modules (BatchNorm1d, ReLU) are applied to
``.features`` in place when the input is sparse and has at least one row.
Call sites: /root/reference/btcdet/models/backbones_3d/spconv_backbone.py:34-42,106-128,656-700."""
from collections import OrderedDict

from torch import nn

from .tensor import SparseConvTensor


import torch as _torch

nn_float32 = _torch.float32
FUSE_CONV_BN = True  # hand the BatchNorm (+ReLU) that follows a sparse conv to the conv's autograd node
FUSE_BN_RELU = True  # set False to run BatchNorm1d / ReLU through torch (used by the parity test)


class SparseModule(nn.Module):
    """marker base class of modules that take a SparseConvTensor"""
    pass


def is_spconv_module(module):
    return isinstance(module, SparseModule)


def is_sparse_conv(module):
    from .conv import SparseConvolution
    return isinstance(module, SparseConvolution)


class SparseSequential(SparseModule):
    def __init__(self, *args, **kwargs):
        super(SparseSequential, self).__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)
        self._sparity_dict = {}

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError('index {} is out of range'.format(idx))
        if idx < 0:
            idx += len(self)
        it = iter(self._modules.values())
        for _ in range(idx):
            next(it)
        return next(it)

    def __len__(self):
        return len(self._modules)

    @property
    def sparity_dict(self):
        return self._sparity_dict

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward_geometry(self, input):
        """the sparse layers' rulebooks only (SparseConvolution.forward_geometry); dense modules are skipped"""
        for module in self._modules.values():
            if is_spconv_module(module):
                input = module.forward_geometry(input)
        return input

    def forward(self, input):
        from . import fused_bn
        from .conv import SparseConvolution
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            module = mods[i]
            i += 1
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                nxt = mods[i] if i < len(mods) else None
                f = input.features
                if FUSE_CONV_BN and FUSE_BN_RELU and nxt is not None and isinstance(module, SparseConvolution) and fused_bn.fusable(nxt) and f.is_cuda \
                        and f.dtype in (nn_float32, _torch.bfloat16) and f.shape[0] != 0:
                    # conv -> BatchNorm1d (-> ReLU): one autograd node, the same kernels (ops.SparseConvBNReLUFunction)
                    relu = i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU
                    input = module(input, fuse_bn=(nxt, relu))
                    i += 1 + int(relu)
                else:
                    input = module(input)
            else:
                if isinstance(input, SparseConvTensor):
                    if input.indices.shape[0] != 0:
                        f = input.features
                        if FUSE_BN_RELU and fused_bn.fusable(module) and f.is_cuda and f.dtype in (nn_float32, _torch.bfloat16) and f.dim() == 2:
                            # BatchNorm1d (+ the ReLU that follows it): one fused HIP call, same parameters / buffers
                            relu = i < len(mods) and type(mods[i]) is nn.ReLU
                            input.features = fused_bn.batch_norm_relu(module, f, relu)
                            i += int(relu)
                        else:
                            input.features = module(f)
                else:
                    input = module(input)
        return input
