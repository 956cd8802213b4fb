"""btcdet_amd -- MI355X (gfx950) implementation of BtcDet's data-parallel hot path.

Voxelizer, occupancy/occlusion target generator, sparse-3D-conv rulebook + fused gather-GEMM-scatter
(forward and backward), exposed through the reference's own operator / module API:

* ``btcdet_amd.spconv``         -- drop-in for the ``spconv`` v1.2.1 surface BtcDet uses
* ``btcdet_amd.processor``      -- ``DataProcessor`` voxelization steps on the GPU
* ``btcdet_amd.occ_targets``    -- ``OccTargets3D`` (occupancy / occlusion grid generator)
* ``btcdet_amd.vfe`` / ``backbones_3d`` / ``occ_head`` / ``pass_occ_vox`` / ``height_compression``

All compute goes through ``libbtcdet_hip.so`` (hand-written HIP, C ABI in ``include/btcdet_hip.h``).
"""
__version__ = "0.1.0"


def install_as_spconv():
    """Register ``btcdet_amd.spconv`` under the module name ``spconv`` so that the reference's
    ``import spconv`` / ``from spconv.utils import VoxelGeneratorV2`` resolve to this implementation
    (/root/reference/btcdet/models/backbones_3d/spconv_backbone.py:3, data_processor.py:64)."""
    import sys
    from . import spconv as _sp
    sys.modules["spconv"] = _sp
    sys.modules["spconv.utils"] = _sp.utils
    sys.modules["spconv.ops"] = _sp.ops
    return _sp
