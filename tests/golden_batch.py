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


FULL_TAGS = ("full_a", "full_b", "full_c", "full_d")


def golden_batch_full(tag, device="cpu"):
    """Rebuilds the batch of tests/golden/btc_<tag>.npz (tests/golden/gen_golden_full.py): the raw scans are regenerated
    (common.raw_scene, checked against the stored SHA-1), range-masked and permuted with the STORED shuffle index, voxelized
    with the CPU oracle -- every intermediate is asserted equal to what the reference's own DataProcessor produced (coords
    and counts in full, payloads by SHA-1) -- and joined in collate_batch + load_data_to_gpu layout.
    -> (g, per-scene dicts as DataProcessor.forward returns them, batch dict of float32 tensors)"""
    import ast
    from oracle import oracle as orc
    g = common.load(tag)
    specs = [dict(ast.literal_eval(str(s))) for s in g["meta_specs"]]
    with_bm = bool(g["meta_with_bm_key"])
    occ = orc.VoxelGeneratorV2(synth.KITTI_OCC_VOXEL, synth.KITTI_OCC_RANGE, 12, 20000)
    det = orc.VoxelGeneratorV2(synth.KITTI_DET_VOXEL, synth.KITTI_DET_RANGE, 5, 16000)
    scenes = []
    for i, spec in enumerate(specs):
        s, raw, raw_pre, bm = common.raw_scene(spec)
        assert np.array_equal(common.sha1(raw), g["raw%d_points_sha1" % i]), "synthetic scene generator drifted"
        keep = orc.mask_points_by_range(raw, synth.KITTI_DET_RANGE)
        pts, pre = raw[keep][g["proc%d_shuffle_idx" % i]], raw_pre[keep]
        assert np.array_equal(common.sha1(pts), g["proc%d_points_sha1" % i])
        r = occ.generate(orc.absxyz_2_cylinxyz_np(pre))
        v = r["voxels"].copy()
        v[..., 1] = v[..., 1] - s["rot_z"]
        d = det.generate(pts)
        assert np.array_equal(r["coordinates"], g["proc%d_voxel_coords" % i]) and np.array_equal(d["coordinates"], g["proc%d_det_voxel_coords" % i])
        assert np.array_equal(r["num_points_per_voxel"], g["proc%d_voxel_num_points" % i])
        assert np.array_equal(d["num_points_per_voxel"], g["proc%d_det_voxel_num_points" % i])
        assert np.array_equal(common.sha1(v), g["proc%d_voxels_sha1" % i]) and np.array_equal(common.sha1(d["voxels"]), g["proc%d_det_voxels_sha1" % i])
        sd = {"raw_points": raw, "raw_pre_rot_points": raw_pre, "points": pts, "masked_pre_rot_points": pre,
              "voxels": v, "voxel_coords": r["coordinates"], "voxel_num_points": r["num_points_per_voxel"],
              "det_voxels": d["voxels"], "det_voxel_coords": d["coordinates"], "det_voxel_num_points": d["num_points_per_voxel"],
              "gt_boxes": g["proc%d_gt_boxes" % i], "box_mirr_flag": np.ones((g["proc%d_gt_boxes" % i].shape[0],), np.float32),
              "rot_z": s["rot_z"], "use_lead_xyz": True, "is_train": True}
        if with_bm:
            sd["bm_points"] = bm
        scenes.append(sd)
    from btcdet_amd.collate import collate_batch, load_data_to_gpu
    keys = ["points", "voxels", "voxel_coords", "voxel_num_points", "det_voxels", "det_voxel_coords", "det_voxel_num_points", "gt_boxes",
            "box_mirr_flag", "rot_z", "use_lead_xyz", "is_train"] + (["bm_points"] if with_bm else [])
    bd = collate_batch([{k: s[k] for k in keys} for s in scenes])
    host = dict(bd)
    bd = load_data_to_gpu(bd, device=device)
    bd["use_occ_prob"] = np.array([True] * bd["batch_size"])
    return g, scenes, bd, host
