"""DataProcessor: the reference's pre-processing queue with the voxelizers (and, on the resident path, the range mask and
the shuffle) on the GPU.

Interface of /root/reference/btcdet/datasets/processor/data_processor.py:7-258 as its callers use it (dataset.py:34-41,
147-152): ``DataProcessor(processor_configs, point_cloud_range, training, occ_config=, det_point_cloud_range=)``, the
attributes ``occ_grid_size / occ_voxel_size / det_grid_size / det_voxel_size / occ_dim / data_processor_queue`` and
``forward(data_dict)``; one step per ``NAME`` in the config list.  Entry points:
  * ``forward(data_dict)``          -- the reference protocol (one scene, numpy in / numpy out) for DataLoader-style callers;
  * ``mask_and_shuffle_batch(...)`` -- SURVEY §8 a1 / a2 for a whole batch in HBM: range mask + stable compaction of
                                       ``points`` and ``pre_rot_points`` (btc_range_mask_compact), then the per-scene
                                       permutation of ``points`` only, as the reference does (btc_gather_rows);
  * ``forward_batch(...)``          -- both voxelizations of a batch that HAS been masked and shuffled (by the call above or
                                       by the caller), no host round trip except one read-back of the two voxel counts;
  * ``forward_raw_batch(...)``      -- the two chained: raw resident points in, voxel keys out.
"""
import numpy as np
import torch

from ._lib import check, f32p, lib, ptr, stream_ptr, workspace
from .spconv.utils import VoxelGeneratorV2


def mask_points_by_range(points, limit_range):
    """bool (N,): x and y inside [lo, hi] (z is not tested) -- common_utils.py:59-62"""
    x, y = points[:, 0], points[:, 1]
    return (x >= limit_range[0]) & (x <= limit_range[3]) & (y >= limit_range[1]) & (y <= limit_range[4])


_UNIT_CORNERS = 0.5 * np.array([[sx, sy, sz] for sz in (-1, 1) for sx, sy in ((1, 1), (1, -1), (-1, -1), (-1, 1))], dtype=np.float64)


def boxes_to_corners_3d(boxes3d):
    """(N,7+) [x,y,z,dx,dy,dz,heading] -> (N,8,3), corner order of OpenPCDet's box_utils.boxes_to_corners_3d"""
    boxes3d = np.asarray(boxes3d)
    local = boxes3d[:, None, 3:6] * _UNIT_CORNERS[None]
    c, s = np.cos(boxes3d[:, 6]), np.sin(boxes3d[:, 6])
    x = local[..., 0] * c[:, None] - local[..., 1] * s[:, None]
    y = local[..., 0] * s[:, None] + local[..., 1] * c[:, None]
    return np.stack([x, y, local[..., 2]], axis=-1) + boxes3d[:, None, 0:3]


def mask_boxes_outside_range_numpy(boxes, limit_range, min_num_corners=1):
    """keep a box if at least min_num_corners of its corners lie inside the range (box_utils.mask_boxes_outside_range_numpy)"""
    corners = boxes_to_corners_3d(np.asarray(boxes)[:, :7])
    inside = ((corners >= limit_range[0:3]) & (corners <= limit_range[3:6])).all(axis=2)
    return inside.sum(axis=1) >= min_num_corners


def cart_to_occ_coords(points, coord_type):
    """GPU absxyz_2_cylinxyz_np / absxyz_2_spherexyz_np (coords_utils.py:268-292)"""
    points = points.contiguous()
    out = torch.empty_like(points)
    mode = {"cylinder": 1, "sphere": 2}[coord_type]
    check(lib().btc_cart_to_occ_coords(ptr(points), ptr(out), points.shape[0], points.shape[1], mode, stream_ptr()),
          "btc_cart_to_occ_coords")
    return out


class DataProcessor(object):
    def __init__(self, processor_configs, point_cloud_range, training, occ_config=None, det_point_cloud_range=None, **kwargs):
        self.point_cloud_range = point_cloud_range
        self.det_point_cloud_range = det_point_cloud_range
        self.occ_config = occ_config
        self.training = training
        self.mode = "train" if training else "test"
        self.occ_dim = None
        self._occ_gen = self._det_gen = None
        builders = {"mask_points_and_boxes_outside_range": self._build_range_mask, "shuffle_points": self._build_shuffle,
                    "transform_points_to_sphere_voxels": self._build_occ_voxels, "det_transform_points_to_voxels": self._build_det_voxels,
                    "transform_points_to_voxels": self._build_single_voxels}
        self.data_processor_queue = []
        for cfg in processor_configs:
            if cfg.NAME not in builders:
                raise NotImplementedError("DataProcessor step %r" % cfg.NAME)
            self.data_processor_queue.append(builders[cfg.NAME](cfg))

    def forward(self, data_dict):
        for step in self.data_processor_queue:
            data_dict = step(data_dict=data_dict)
        return data_dict

    # ------------------------------------------------------------------ steps of the reference protocol (per scene, numpy)
    def _build_range_mask(self, cfg):
        drop_boxes = bool(cfg.REMOVE_OUTSIDE_BOXES) and self.training
        corners_needed = cfg.get("min_num_corners", 1)

        def step(data_dict):
            keep = mask_points_by_range(data_dict["points"], self.det_point_cloud_range)
            for key in ("points", "pre_rot_points"):            # the un-rotated copy follows the same mask (:27-28)
                if key in data_dict:
                    data_dict[key] = data_dict[key][keep]
            boxes = data_dict.get("gt_boxes", None)
            if boxes is not None and drop_boxes:
                data_dict["gt_boxes"] = boxes[mask_boxes_outside_range_numpy(boxes, self.det_point_cloud_range, corners_needed)]
            return data_dict
        return step

    def _build_shuffle(self, cfg):
        enabled = self._shuffle_flag = bool(cfg.SHUFFLE_ENABLED[self.mode])

        def step(data_dict):
            if enabled:   # the global numpy RNG, as in the reference (tools/train.py seeds it); `points` only (:41-51)
                data_dict["points"] = data_dict["points"][np.random.permutation(data_dict["points"].shape[0])]
            return data_dict
        return step

    def _make_gen(self, config, rng):
        gen = VoxelGeneratorV2(voxel_size=config.VOXEL_SIZE, point_cloud_range=rng, max_num_points=config.MAX_POINTS_PER_VOXEL,
                               max_voxels=config.MAX_NUMBER_OF_VOXELS[self.mode])
        grid = (np.asarray(rng[3:6]) - np.asarray(rng[0:3])) / np.array(config.VOXEL_SIZE)
        return gen, np.round(grid).astype(np.int64)

    @staticmethod
    def _store(data_dict, prefix, out, use_lead_xyz):
        voxels = out["voxels"] if use_lead_xyz else out["voxels"][..., 3:]
        data_dict[prefix + "voxels"], data_dict[prefix + "voxel_coords"], data_dict[prefix + "voxel_num_points"] = \
            voxels, out["coordinates"], out["num_points_per_voxel"]
        return data_dict

    def _build_occ_voxels(self, cfg):
        self._occ_gen, self.occ_grid_size = self._make_gen(cfg, self.point_cloud_range)
        self.occ_voxel_size = cfg.VOXEL_SIZE
        self.max_points_per_voxel = cfg.MAX_POINTS_PER_VOXEL

        def step(data_dict):
            unrotated = "pre_rot_points" in data_dict
            src = data_dict["pre_rot_points"] if unrotated else data_dict["points"]
            pts = torch.from_numpy(np.ascontiguousarray(src, dtype=np.float32)).cuda()
            out = self._occ_gen.generate(cart_to_occ_coords(pts, self.occ_config.COORD_TYPE))
            out = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in out.items()}
            if not data_dict["use_lead_xyz"]:   # the xyz columns go first (:141-142) ...
                out["voxels"] = out["voxels"][..., 3:]
            if unrotated:   # ... then column 1 of WHAT IS LEFT moves by the scene's rotation, padded slots too (:148-149)
                data_dict.pop("pre_rot_points")
                out["voxels"][..., 1] = out["voxels"][..., 1] - np.float32(data_dict["rot_z"])
            return self._store(data_dict, "", out, True)
        return step

    def _build_det_voxels(self, cfg):
        self._det_gen, self.det_grid_size = self._make_gen(cfg, self.det_point_cloud_range)
        self.det_voxel_size = cfg.VOXEL_SIZE

        def step(data_dict):
            out = self._det_gen.generate(np.ascontiguousarray(data_dict["points"], dtype=np.float32))
            return self._store(data_dict, "det_", out, data_dict["use_lead_xyz"])
        return step

    def _build_single_voxels(self, cfg):
        gen, grid = self._make_gen(cfg, self.point_cloud_range)
        self.occ_grid_size = self.det_grid_size = grid
        self.occ_voxel_size = self.det_voxel_size = cfg.VOXEL_SIZE

        def step(data_dict):
            out = gen.generate(np.ascontiguousarray(data_dict["points"], dtype=np.float32))
            return self._store(data_dict, "", out, data_dict["use_lead_xyz"])
        return step

    # ------------------------------------------------------------------ resident path (whole batch in HBM)
    def mask_and_shuffle_batch(self, points, pre_rot_points, scene_offsets, shuffle_idx=None, generator=None):
        """SURVEY §8 a1 + a2 on the device.  points / pre_rot_points (sum N, C) f32 raw scans (scenes contiguous; pre_rot_points
        may be None), scene_offsets (B+1) i32.
        -> (points', pre_rot_points', scene_offsets' (device i32), per-scene counts (host list)).
        Range mask: both arrays, same mask, order kept.  Shuffle (training mode with SHUFFLE_ENABLED): `points` only, like
        the reference.  shuffle_idx: list of B integer arrays, the permutations to apply -- hand in what
        ``np.random.permutation(n_b)`` returned to consume the reference's RNG stream (tools/train.py seeds it); None draws
        ``torch.randperm`` on the device (generator optional), which is NOT that stream.  One (B+1)-int read-back: the
        per-scene counts size the outputs (and the permutations)."""
        L, dev = lib(), points.device
        points = points.contiguous()
        n, ld = points.shape
        B = scene_offsets.numel() - 1
        offs = scene_offsets.to(device=dev, dtype=torch.int32).contiguous()
        out = torch.empty_like(points)
        pre = out_pre = None
        if pre_rot_points is not None:
            pre = pre_rot_points.contiguous()
            out_pre = torch.empty_like(pre)
        new_offs = torch.empty((B + 1,), dtype=torch.int32, device=dev)
        rng = np.ascontiguousarray(np.asarray(self.det_point_cloud_range, dtype=np.float32)[[0, 1, 3, 4]])
        ws_bytes = L.btc_range_mask_ws_bytes(n)
        ws = workspace(ws_bytes, dev)
        check(L.btc_range_mask_compact(ptr(points), ptr(pre), n, ld, pre.shape[1] if pre is not None else 0, ptr(offs), B, f32p(rng),
                                       ptr(out), ptr(out_pre), ptr(new_offs), None, ptr(ws), ws_bytes, stream_ptr()),
              "btc_range_mask_compact")
        bounds = new_offs.tolist()
        counts = [bounds[b + 1] - bounds[b] for b in range(B)]
        out = out[:bounds[B]]
        if out_pre is not None:
            out_pre = out_pre[:bounds[B]]
        if self._shuffle_enabled():
            if shuffle_idx is None:
                parts = [torch.randperm(c, device=dev, generator=generator).to(torch.int32) + bounds[b] for b, c in enumerate(counts)]
            else:
                assert len(shuffle_idx) == B and all(len(p) == c for p, c in zip(shuffle_idx, counts)), "one permutation per scene"
                parts = [torch.as_tensor(np.asarray(p), device=dev).to(torch.int32) + bounds[b] for b, p in enumerate(shuffle_idx)]
            idx = torch.cat(parts) if parts else torch.zeros((0,), dtype=torch.int32, device=dev)
            shuffled = torch.empty_like(out)
            bad = torch.zeros((1,), dtype=torch.int32, device=dev)
            check(L.btc_gather_rows(ptr(out), ptr(idx.contiguous()), out.shape[0], ld, out.shape[0], ptr(shuffled), ptr(bad), stream_ptr()),
                  "btc_gather_rows")
            out = shuffled
        return out, out_pre, new_offs, counts

    def _shuffle_enabled(self):
        flag = getattr(self, "_shuffle_flag", None)
        return bool(flag) if flag is not None else False

    def forward_raw_batch(self, points, pre_rot_points, scene_offsets, rot_z, shuffle_idx=None, generator=None):
        """raw resident scans -> the voxel keys of collate_batch: mask_and_shuffle_batch, then forward_batch"""
        pts, pre, offs, counts = self.mask_and_shuffle_batch(points, pre_rot_points, scene_offsets, shuffle_idx, generator)
        out = self.forward_batch(pts, pre if pre is not None else pts, offs, rot_z)
        out.update({"masked_points": pts, "masked_pre_rot_points": pre, "scene_offsets": offs, "scene_counts": counts})
        return out

    def forward_batch(self, points, pre_rot_points, scene_offsets, rot_z):
        """points / pre_rot_points (sum N, 4) f32 on the GPU, ALREADY range-masked and shuffled (mask_and_shuffle_batch does
        both; this function does neither), scenes contiguous; scene_offsets (B+1) i32, rot_z (B) f32
        -> the voxel keys of collate_batch (dataset.py:185-192), on the GPU."""
        cyl = cart_to_occ_coords(pre_rot_points, self.occ_config.COORD_TYPE)
        # both voxelizations are enqueued back to back; ONE read-back returns the two voxel counts
        totals = torch.empty((2,), dtype=torch.int32, device=points.device)     # (one tensor for both counts: no cat in front of the read-back)
        vox, coords, num, _ = self._occ_gen.generate_batch(cyl, scene_offsets, sync=False, total_out=totals[0:1])
        dvox, dcoords, dnum, _ = self._det_gen.generate_batch(points, scene_offsets, sync=False, total_out=totals[1:2])
        m_occ, m_det = totals.tolist()
        vox, coords, num = vox[:m_occ], coords[:m_occ], num[:m_occ]
        dvox, dcoords, dnum = dvox[:m_det], dcoords[:m_det], dnum[:m_det]
        if m_occ > 0:
            check(lib().btc_voxel_shift_col(ptr(vox), ptr(coords), m_occ, vox.shape[1], vox.shape[2], 1, ptr(rot_z), -1.0,
                                            stream_ptr()), "btc_voxel_shift_col")
        # "__voxels_owned__": `voxels` is a fresh tensor nobody else holds -- OccTargets3D may write the absolute coordinates into it in place
        return {"voxels": vox, "voxel_coords": coords, "voxel_num_points": num, "det_voxels": dvox,
                "det_voxel_coords": dcoords, "det_voxel_num_points": dnum, "__voxels_owned__": True}
