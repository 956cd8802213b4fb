"""Ground-truth database sampling ("gt_sampling"): pastes objects cut out of other frames into the scene before it is
voxelized -- a data-side producer of the hot path's inputs.

Same constructor / call protocol and semantics as /root/reference/btcdet/datasets/augmentor/database_sampler.py:8-217
(``DataBaseSampler(root_path, sampler_cfg, class_names, db_infos, logger)``; ``sampler(data_dict)``): PREPARE filters, the
per-class pointer into a permutation drawn from the global numpy RNG, rejection of candidates whose BEV footprint overlaps an
existing box or another candidate (rotated BEV IoU through the HIP kernel, btcdet_amd.iou3d_nms), removal of scene points inside
the enlarged pasted boxes, optional road-plane snapping, and the keys it writes (``gt_boxes, gt_names, points, gt_boxes_inds,
augment_box_num, aug_boxes_image_idx, aug_boxes_gt_idx``; ``gt_boxes_mask`` is consumed).  Checked against the reference's
own class run over restated geometric primitives (tests/golden/gen_data_side_golden.py, tests/test_hip_database_sampler.py)."""
import pathlib

import numpy as np

from . import iou3d_nms


def points_in_boxes_mask(points_xyz, boxes, margin=1e-2):
    """(N, M) bool: point m inside box n -- the CPU test of roiaware_pool3d (src/roiaware_pool3d.cpp:128-140): |z - cz| <= dz/2,
    and the point rotated into the box frame within dx/2 + margin, dy/2 + margin (strict)"""
    p = np.asarray(points_xyz, dtype=np.float32)
    b = np.asarray(boxes, dtype=np.float32)
    sx, sy = p[None, :, 0] - b[:, None, 0], p[None, :, 1] - b[:, None, 1]
    c, s = np.cos(-b[:, 6]).astype(np.float32)[:, None], np.sin(-b[:, 6]).astype(np.float32)[:, None]
    lx, ly = sx * c - sy * s, sx * s + sy * c
    inside_z = np.abs(p[None, :, 2] - b[:, None, 2]) <= b[:, None, 5] / np.float32(2.0)
    return inside_z & (np.abs(lx) < b[:, None, 3] / np.float32(2.0) + np.float32(margin)) & (np.abs(ly) < b[:, None, 4] / np.float32(2.0) + np.float32(margin))


class _ClassCursor(object):
    """walks one class's database entries in a random order, `count` at a time, reshuffling when the order is used up"""

    def __init__(self, count, n_entries):
        self.count = count          # kept as given (a string in the reference's cfg: 'Car:15')
        self.pointer = n_entries    # forces a permutation on first use
        self.indices = np.arange(n_entries)

    def take(self, entries):
        n = int(self.count)
        if self.pointer >= len(entries):
            self.indices = np.random.permutation(len(entries))
            self.pointer = 0
        picked = [entries[i] for i in self.indices[self.pointer:self.pointer + n]]
        self.pointer += n
        return picked


class DataBaseSampler(object):
    def __init__(self, root_path, sampler_cfg, class_names, db_infos, logger=None):
        self.root_path = pathlib.Path(root_path)
        self.sampler_cfg, self.class_names, self.logger = sampler_cfg, class_names, logger
        self.db_infos = db_infos
        for step, arg in sampler_cfg.PREPARE.items():
            self.db_infos = {"filter_by_difficulty": self.filter_by_difficulty, "filter_by_min_points": self.filter_by_min_points}[step](self.db_infos, arg)
        self.limit_whole_scene = sampler_cfg.get("LIMIT_WHOLE_SCENE", False)
        self.sample_class_num, self.sample_groups = {}, {}
        for spec in sampler_cfg.SAMPLE_GROUPS:
            name, num = spec.split(":")
            if name in class_names:
                self.sample_class_num[name] = num
                self.sample_groups[name] = _ClassCursor(num, len(self.db_infos[name]))

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k != "logger"}

    def __setstate__(self, d):
        self.__dict__.update(d)
        self.logger = None

    def filter_by_difficulty(self, db_infos, removed_difficulty):
        return {name: [e for e in entries if e["difficulty"] not in removed_difficulty] for name, entries in db_infos.items()}

    def filter_by_min_points(self, db_infos, min_gt_points_list):
        for spec in min_gt_points_list:
            name, least = spec.split(":")
            if int(least) > 0 and name in db_infos:
                db_infos[name] = [e for e in db_infos[name] if e["num_points_in_gt"] >= int(least)]
        return db_infos

    @staticmethod
    def put_boxes_on_road_planes(gt_boxes, road_planes, calib):
        """KITTI only: drop the boxes onto the plane a x + b y + c z + d = 0 given in camera coordinates"""
        a, b, c, d = road_planes
        cam = calib.lidar_to_rect(gt_boxes[:, 0:3])
        cam[:, 1] = (-d - a * cam[:, 0] - c * cam[:, 2]) / b
        lidar_h = calib.rect_to_lidar(cam)[:, 2]
        lift = gt_boxes[:, 2] - gt_boxes[:, 5] / 2 - lidar_h
        gt_boxes[:, 2] -= lift
        return gt_boxes, lift

    def _paste(self, data_dict, new_boxes, new_entries):
        keep = data_dict["gt_boxes_mask"]
        boxes, names = data_dict["gt_boxes"][keep], data_dict["gt_names"][keep]
        data_dict["gt_boxes_inds"] = data_dict["gt_boxes_inds"][keep]
        lift = None
        if self.sampler_cfg.get("USE_ROAD_PLANE", False):
            new_boxes, lift = self.put_boxes_on_road_planes(new_boxes, data_dict["road_plane"], data_dict["calib"])
            data_dict.pop("calib")
            data_dict.pop("road_plane")
        clouds = []
        for i, e in enumerate(new_entries):
            obj = np.fromfile(str(self.root_path / e["path"]), dtype=np.float32).reshape([-1, self.sampler_cfg.NUM_POINT_FEATURES])
            obj[:, :3] += e["box3d_lidar"][:3]
            if lift is not None:
                obj[:, 2] -= lift[i]
            clouds.append(obj)
        grown = np.array(new_boxes[:, 0:7], dtype=np.float32, copy=True)
        extra = self.sampler_cfg.REMOVE_EXTRA_WIDTH
        if sum(extra) > 1e-3:
            grown[:, 3:6] += np.asarray(extra, dtype=np.float32)[None, :]
        scene = data_dict["points"]
        scene = scene[~points_in_boxes_mask(scene[:, 0:3], grown).any(axis=0)]
        data_dict["points"] = np.concatenate([scene] + clouds, axis=0)
        new_names = np.array([e["name"] for e in new_entries])
        if boxes.ndim != 2 or boxes.shape[0] == 0:
            data_dict["gt_boxes"], data_dict["gt_names"] = new_boxes, new_names
        else:
            data_dict["gt_boxes"], data_dict["gt_names"] = np.concatenate([boxes, new_boxes], axis=0), np.concatenate([names, new_names], axis=0)
        data_dict["augment_box_num"] = new_boxes.shape[0]
        return data_dict

    def __call__(self, data_dict):
        gt_boxes, gt_names = data_dict["gt_boxes"], data_dict["gt_names"].astype(str)
        placed = gt_boxes
        entries, image_idx, gt_idx = [], [], []
        for name, cursor in self.sample_groups.items():
            if self.limit_whole_scene:
                cursor.count = str(int(self.sample_class_num[name]) - int(np.sum(name == gt_names)))
            if int(cursor.count) <= 0:
                continue
            cand = cursor.take(self.db_infos[name])
            cboxes = np.stack([e["box3d_lidar"] for e in cand], axis=0).astype(np.float32)
            cimg = np.stack([e["image_idx"] if "image_idx" in e else e["sample_idx"] for e in cand], axis=0).astype(np.int32)
            cgt = np.stack([e["gt_idx"] for e in cand], axis=0).astype(np.int32)
            if self.sampler_cfg.get("DATABASE_WITH_FAKELIDAR", False):
                cboxes = _fakelidar_to_lidar(cboxes)
            among = iou3d_nms.boxes_bev_iou_cpu(cboxes[:, 0:7], cboxes[:, 0:7])
            among[range(len(cand)), range(len(cand))] = 0
            versus = among
            if placed.ndim == 2 and placed.shape[0] > 0:
                versus = iou3d_nms.boxes_bev_iou_cpu(cboxes[:, 0:7], placed[:, 0:7])
            ok = np.nonzero((versus.max(axis=1) + among.max(axis=1)) == 0)[0]
            placed = cboxes[ok] if (placed.ndim != 2 or placed.shape[0] == 0) else np.concatenate((placed, cboxes[ok]), axis=0)
            entries.extend(cand[i] for i in ok)
            image_idx.append(cimg[ok])
            gt_idx.append(cgt[ok])
        if entries:
            data_dict = self._paste(data_dict, placed[gt_boxes.shape[0]:, :], entries)
            data_dict["aug_boxes_image_idx"] = np.concatenate(image_idx, axis=0)
            data_dict["aug_boxes_gt_idx"] = np.concatenate(gt_idx, axis=0)
        data_dict.pop("gt_boxes_mask")
        return data_dict


def _fakelidar_to_lidar(boxes):
    """box_utils.boxes3d_kitti_fakelidar_to_lidar: [x, y, z (bottom), w, l, h, r] -> [x, y, z (centre), l, w, h, -(r + pi/2)]"""
    out = np.array(boxes, copy=True)
    w, l, h, r = boxes[:, 3:4], boxes[:, 4:5], boxes[:, 5:6], boxes[:, 6:7]
    out[:, 2] += h[:, 0] / 2
    return np.concatenate([out[:, 0:3], l, w, h, -(r + np.pi / 2)], axis=-1)
