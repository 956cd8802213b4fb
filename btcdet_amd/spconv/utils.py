"""spconv.utils replacement: VoxelGeneratorV2 / VoxelGenerator on the GPU.

Constructor and ``generate`` contract as used at
/root/reference/btcdet/datasets/processor/data_processor.py:63-73,85-90,107-117,136-141,160-182
(kwargs ``voxel_size, point_cloud_range, max_num_points, max_voxels``; ``generate(points)`` returns a
dict with ``voxels``, ``coordinates``, ``num_points_per_voxel``).

``generate`` accepts a numpy array (the reference's DataLoader path: host in, host out) or a GPU
tensor (resident path: stays on the device).  ``generate_batch`` voxelizes a whole collated batch in
one launch sequence and is what the resident training step uses.
"""
import numpy as np
import torch

from .. import _lib
from .._lib import check, f32p, i3p, lib, ptr, stream_ptr, workspace


def voxelize_batch(points, scene_offsets, point_cloud_range, voxel_size, grid_size, max_points, max_voxels,
                   xyz_col=0, feat_col=0, num_feat=None, sync=True, total_out=None):
    """points (n, ld) float32 GPU tensor with scenes stored contiguously; scene_offsets (B+1) int32
    GPU tensor.  Returns voxels (M,P,C), coords (M,4) [b,z,y,x] int32, num (M,) int32.
    With sync=False the tensors keep their capacity (B*max_voxels rows) and the voxel count stays on
    the device as the 4th return value (total_out: a caller's (1,) int32 device tensor to hold it -- two voxelizations whose counts
    come back in ONE read-back write into the halves of one tensor)."""
    if not points.is_cuda:
        raise _lib.BtcHipError("voxelize_batch: points must live on the GPU")
    points = points.contiguous()
    if points.dtype != torch.float32:
        raise _lib.BtcHipError("voxelize_batch: float32 points expected")
    n, ld = points.shape
    C = int(num_feat if num_feat is not None else ld - feat_col)
    batch = scene_offsets.numel() - 1
    rng = np.ascontiguousarray(np.asarray(point_cloud_range, dtype=np.float32))
    vs = np.ascontiguousarray(np.asarray(voxel_size, dtype=np.float32))
    grid = np.ascontiguousarray(np.asarray(grid_size, dtype=np.int32))
    dev = points.device
    cap = max(min(batch * int(max_voxels), n), 1)
    voxels = torch.empty((cap, int(max_points), C), dtype=torch.float32, device=dev)
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    num = torch.empty((cap,), dtype=torch.int32, device=dev)
    d_total = total_out if total_out is not None else torch.empty((1,), dtype=torch.int32, device=dev)     # (written by every path of btc_voxelize)
    L = lib()
    ws_bytes = L.btc_voxelize_ws_bytes(n, batch, int(max_points))
    ws = workspace(ws_bytes, dev)
    offs = scene_offsets.to(device=dev, dtype=torch.int32).contiguous()
    check(L.btc_voxelize(ptr(points), n, ld, int(xyz_col), int(feat_col), C, ptr(offs), batch, f32p(rng), f32p(vs),
                         i3p(grid), int(max_points), int(max_voxels), ptr(voxels), ptr(coords), ptr(num), ptr(d_total),
                         ptr(ws), ws_bytes, stream_ptr()), "btc_voxelize")
    if not sync:
        return voxels, coords, num, d_total
    m = int(d_total.item())
    return voxels[:m], coords[:m], num[:m]


class VoxelGeneratorV2(object):
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000, full_mean=False,
                 block_filtering=False, block_factor=8, block_size=3, height_threshold=0.1, height_high_threshold=2.0):
        assert not full_mean and not block_filtering, "full_mean / block_filtering are not used by BtcDet"
        point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        voxel_size = np.array(voxel_size, dtype=np.float32)
        grid_size = (point_cloud_range[3:] - point_cloud_range[:3]) / voxel_size
        grid_size = np.round(grid_size).astype(np.int64)
        self._voxel_size = voxel_size
        self._point_cloud_range = point_cloud_range
        self._max_num_points = int(max_num_points)
        self._max_voxels = int(max_voxels)
        self._grid_size = grid_size

    def generate(self, points, max_voxels=None):
        mv = int(max_voxels or self._max_voxels)
        is_np = isinstance(points, np.ndarray)
        if is_np:
            if not torch.cuda.is_available():
                raise _lib.BtcHipError("VoxelGeneratorV2.generate needs the GPU (no CPU fallback)")
            pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).cuda()
        else:
            pts = points
        offs = torch.tensor([0, pts.shape[0]], dtype=torch.int32, device=pts.device)
        voxels, coords, num = voxelize_batch(pts, offs, self._point_cloud_range, self._voxel_size, self._grid_size,
                                             self._max_num_points, mv)
        coords = coords[:, 1:].contiguous()
        res = {"voxels": voxels, "coordinates": coords, "num_points_per_voxel": num, "voxel_num": voxels.shape[0]}
        if is_np:
            res = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in res.items()}
        return res

    def generate_batch(self, points, scene_offsets, xyz_col=0, feat_col=0, num_feat=None, max_voxels=None, sync=True, total_out=None):
        return voxelize_batch(points, scene_offsets, self._point_cloud_range, self._voxel_size, self._grid_size,
                              self._max_num_points, int(max_voxels or self._max_voxels), xyz_col, feat_col, num_feat, sync, total_out)

    @property
    def voxel_size(self):
        return self._voxel_size

    @property
    def max_num_points_per_voxel(self):
        return self._max_num_points

    @property
    def point_cloud_range(self):
        return self._point_cloud_range

    @property
    def grid_size(self):
        return self._grid_size


VoxelGenerator = VoxelGeneratorV2
