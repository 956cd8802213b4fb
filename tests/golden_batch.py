"""Rebuilds the reference-side batch_dict of tests/golden/btc_small.npz (collate_batch +
load_data_to_gpu layout: everything float32) from the stored per-scene processor outputs."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import common  # noqa: E402

from btcdet_amd import synth  # noqa: E402


def golden_batch(tag="small", device="cpu"):
    g = common.load(tag)
    seeds, az = [int(s) for s in g["meta_seeds"]], float(g["meta_az_step"])
    scenes = [synth.make_scene(s, az_step=az) for s in seeds]
    B = len(scenes)
    pad = lambda a, i: np.pad(a, ((0, 0), (1, 0)), mode="constant", constant_values=i)
    cat = lambda k: np.concatenate([g["proc%d_%s" % (i, k)] for i in range(B)], axis=0)
    catp = lambda k: np.concatenate([pad(g["proc%d_%s" % (i, k)], i) for i in range(B)], axis=0)
    gts = [g["proc%d_gt_boxes" % i] for i in range(B)]
    maxg = max(len(x) for x in gts)
    gt = np.zeros((B, maxg, 8), np.float32)
    mirr = np.zeros((B, maxg), np.float32)
    for i, x in enumerate(gts):
        gt[i, :len(x)] = x
        mirr[i, :len(x)] = 1.0
    bd = {
        "voxels": cat("voxels"), "voxel_num_points": cat("voxel_num_points"), "voxel_coords": catp("voxel_coords"),
        "det_voxels": cat("det_voxels"), "det_voxel_num_points": cat("det_voxel_num_points"),
        "det_voxel_coords": catp("det_voxel_coords"), "points": catp("points"),
        "bm_points": np.concatenate([pad(s["bm_points"], i) for i, s in enumerate(scenes)], axis=0),
        "gt_boxes": gt, "box_mirr_flag": mirr, "rot_z": np.array([s["rot_z"] for s in scenes]),
    }
    bd = {k: torch.from_numpy(np.asarray(v)).float().to(device) for k, v in bd.items()}
    bd.update({"batch_size": B, "gt_boxes_num": [len(x) for x in gts], "is_train": True,
               "use_occ_prob": np.array([True] * B)})
    return g, scenes, bd
