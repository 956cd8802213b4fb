"""PassOccVox: occupancy probabilities -> added occupancy points -> merged detection voxels.

Mirror of /root/reference/btcdet/models/occ_pnt/pass_occ_vox.py:10-59 and
add_occ_template.py:78-190,248-268 (module protocol ``forward(batch_dict) -> batch_dict``).
The GPU re-voxelization (torch.unique(dim=0) + sort + scatter-pad + .cpu() sync in the reference)
is one HIP call pair (btc_revoxelize_count / btc_revoxelize_fill).
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, i3, i3p, lib, ptr, stream_ptr, workspace


def revoxelize(points, coords, batch_size, grid_zyx):
    """combine_gt_occ_voxel_point (add_occ_template.py:262-268) on the GPU.
    points (n,C) f32, coords (n,4) int64 [b,z,y,x] -> voxels (M,Pmax,C) f32, num (M,) i64, vcoords (M,4) i64
    (cells ascending in (b,z,y,x), points of a cell in input order, zero padded)."""
    points = points.contiguous()
    coords = coords.contiguous()
    if coords.dtype != torch.int64 or points.dtype != torch.float32:
        raise _lib.BtcHipError("revoxelize: int64 coords and float32 points expected")
    n, C = points.shape
    dev = points.device
    sh = i3([int(v) for v in grid_zyx])
    L = lib()
    ws_bytes = L.btc_revoxelize_ws_bytes(n, int(batch_size), i3p(sh))
    ws = workspace(ws_bytes, dev)
    d_mp = torch.zeros((2,), dtype=torch.int32, device=dev)
    check(L.btc_revoxelize_count(ptr(coords), n, int(batch_size), i3p(sh), ptr(d_mp[0:1]), ptr(d_mp[1:2]), ptr(ws),
                                 ws_bytes, stream_ptr()), "btc_revoxelize_count")
    m, pmax = [int(v) for v in d_mp.tolist()]  # the one read-back (the reference syncs on Pmax too, :251)
    voxels = torch.empty((m, pmax, C), dtype=torch.float32, device=dev)
    vcoords = torch.empty((m, 4), dtype=torch.int64, device=dev)
    vnum = torch.empty((m,), dtype=torch.int64, device=dev)
    check(L.btc_revoxelize_fill(ptr(points), ptr(coords), n, C, int(batch_size), i3p(sh), m, pmax, ptr(voxels),
                                ptr(vcoords), ptr(vnum), ptr(ws), ws_bytes, stream_ptr()), "btc_revoxelize_fill")
    return voxels, vnum, vcoords
