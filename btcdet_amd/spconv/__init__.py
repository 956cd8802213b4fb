"""Drop-in replacement for the ``spconv`` v1.2.1 surface BtcDet uses."""
from . import ops, utils
from .conv import (SparseConv2d, SparseConv3d, SparseConvolution, SparseConvTranspose2d, SparseConvTranspose3d,
                   SparseInverseConv2d, SparseInverseConv3d, SubMConv2d, SubMConv3d)
from .modules import SparseModule, SparseSequential
from .pool import SparseMaxPool2d, SparseMaxPool3d
from .tensor import SparseConvTensor

__all__ = ["ops", "utils", "SparseConv2d", "SparseConv3d", "SparseConvolution", "SparseConvTranspose2d",
           "SparseConvTranspose3d", "SparseInverseConv2d", "SparseInverseConv3d", "SubMConv2d", "SubMConv3d",
           "SparseModule", "SparseSequential", "SparseMaxPool2d", "SparseMaxPool3d", "SparseConvTensor"]
