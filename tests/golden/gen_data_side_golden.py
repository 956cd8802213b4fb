"""Generates tests/golden/data_side.npz by running the REFERENCE's own data-side producers of hot-path inputs (SURVEY.md §8f
row 3) in this container, on seeded inputs and on best-match template files written in the reference's on-disk format:
  * augmentor_utils.global_rotation + DataAugmentor.random_world_rotation with SAVE_PRE_ROT (data_augmentor.py:136-155):
    `points`, `pre_rot_points`, `rot_z`, rotated `gt_boxes` and `bm_points`
  * augmentor_utils.global_scaling / random_flip_along_x (data_augmentor.py:103-134,157-171)
  * MltBestMatchQuerier.__call__ (multi_best_match_querier.py:50-98,269-296): `bm_points` from `{frame}_{box}.pkl` templates
  * KittiDataset.get_lidar's format (kitti_dataset.py:72-75): float32 (N, 4) `.bin`

    python tests/golden/gen_data_side_golden.py
The reference cannot travel to the GPU box; the vectors can."""
import os
import pickle
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
import scipy.spatial  # noqa: E402,F401
import torch  # noqa: E402,F401

np.int = int
np.float = float


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _ED(dict):
    __getattr__ = dict.get
    __setattr__ = dict.__setitem__


_mod("easydict", EasyDict=_ED)
_mod("skimage"); _mod("skimage.io"); _mod("skimage.draw", line_aa=None)
_mod("spconv"); _mod("spconv.utils")
for n in ["btcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda", "btcdet.ops.iou3d_nms.iou3d_nms_cuda"]:
    _mod(n)

from btcdet.datasets.augmentor import augmentor_utils  # noqa: E402
from btcdet.datasets.augmentor.data_augmentor import DataAugmentor  # noqa: E402
from btcdet.datasets.augmentor.multi_best_match_querier import MltBestMatchQuerier  # noqa: E402

rng = np.random.default_rng(77)
out = {}
n_pts, n_box = 600, 5
points = rng.uniform([0, -40, -3, 0], [70, 40, 1, 1], (n_pts, 4)).astype(np.float32)
gt_boxes = np.concatenate([rng.uniform([5, -30, -2], [60, 30, 0], (n_box, 3)), rng.uniform([3.2, 1.4, 1.3], [4.5, 1.9, 1.8], (n_box, 3)),
                           rng.uniform(-3.1, 3.1, (n_box, 1))], axis=1).astype(np.float32)
bm = rng.uniform(-3, 3, (150, 3)).astype(np.float32)
out["points"], out["gt_boxes"], out["bm_in"] = points, gt_boxes, bm

# ---- global_rotation (seeded numpy global RNG, as the reference draws it)
np.random.seed(11)
b, p, noise, sp = augmentor_utils.global_rotation(gt_boxes.copy(), points.copy(), [-0.78539816, 0.78539816], [bm.copy()])
out["rot_boxes"], out["rot_points"], out["rot_noise"], out["rot_bm"] = b, p, np.float64(noise), sp[0]

# ---- DataAugmentor.random_world_rotation with SAVE_PRE_ROT
aug = object.__new__(DataAugmentor)
np.random.seed(12)
d = aug.random_world_rotation({"points": points.copy(), "gt_boxes": gt_boxes.copy(), "bm_points": bm.copy()},
                              config=_ED(WORLD_ROT_ANGLE=[-0.78539816, 0.78539816], SAVE_PRE_ROT=True))
out["wr_points"], out["wr_pre_rot_points"], out["wr_rot_z"], out["wr_gt_boxes"], out["wr_bm"] = \
    d["points"], d["pre_rot_points"], np.float64(d["rot_z"]), d["gt_boxes"], d["bm_points"]

# ---- scaling / flip
np.random.seed(13)
b, p, sp = augmentor_utils.global_scaling(gt_boxes.copy(), points.copy(), [0.95, 1.05], [bm.copy()])
out["sc_boxes"], out["sc_points"], out["sc_bm"] = b, p, sp[0]
b, p, sp = augmentor_utils.random_flip_along_x(gt_boxes.copy(), points.copy(), [bm.copy()], enable=True)
out["fl_boxes"], out["fl_points"], out["fl_bm"] = b, p, sp[0]

# ---- best-match templates on disk -> bm_points
tmp = Path(tempfile.mkdtemp())
(tmp / "bm_car").mkdir()
frame, box_ids, names = 123, np.array([4, 0, 7, 2, 9]), np.array(["Car", "Car", "Pedestrian", "Car", "Car"])
templates = []
for i, bid in enumerate(box_ids):
    t = rng.uniform(-2, 2, (40 + 13 * i, 3)).astype(np.float32)
    templates.append(t)
    if names[i] == "Car":
        with open(tmp / "bm_car" / ("%d_%d.pkl" % (frame, bid)), "wb") as f:
            pickle.dump(t.reshape(-1), f)          # the files hold a flat float array, reshaped by the querier
q = object.__new__(MltBestMatchQuerier)
q.class_names, q.mlt_bm_root, q.load_point_features, q.querier_cfg = ["Car"], {"Car": tmp / "bm_car"}, 3, _ED()
d = q({"gt_boxes": gt_boxes.copy(), "gt_names": names, "gt_boxes_inds": box_ids, "frame_id": "%06d" % frame, "points": points.copy()})
out["bmq_points"] = d["bm_points"]
out["bmq_box_ids"], out["bmq_is_car"] = box_ids, (names == "Car")
for i, t in enumerate(templates):
    out["bmq_template_%d" % i] = t

# ---- KITTI lidar .bin
binf = tmp / "000123.bin"
points.tofile(str(binf))
out["bin_bytes"] = np.frombuffer(open(binf, "rb").read(), dtype=np.uint8)

np.savez_compressed(os.path.join(HERE, "data_side.npz"), **out)
print("wrote", os.path.join(HERE, "data_side.npz"), {k: getattr(v, "shape", None) for k, v in out.items()})
