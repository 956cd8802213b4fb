"""the data-side producers of hot-path inputs (btcdet_amd/data_side.py) against vectors produced by the
reference's own functions (tests/golden/gen_data_side_golden.py ran augmentor_utils.global_rotation / global_scaling /
random_flip_along_x, DataAugmentor.random_world_rotation and MltBestMatchQuerier.__call__ in the build container), and the
two on-disk formats."""
import os
import pickle

import numpy as np
import pytest
import torch

from btcdet_amd import data_side as ds

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data_side.npz"))


def test_global_rotation_matches_reference_bit_for_bit():
    np.random.seed(11)
    b, p, noise, sp = ds.global_rotation(G["gt_boxes"].copy(), G["points"].copy(), [-0.78539816, 0.78539816], [G["bm_in"].copy()])
    assert float(noise) == float(G["rot_noise"])
    np.testing.assert_array_equal(b, G["rot_boxes"])
    np.testing.assert_array_equal(p, G["rot_points"])
    np.testing.assert_array_equal(sp[0], G["rot_bm"])


def test_random_world_rotation_save_pre_rot():
    np.random.seed(12)
    d = ds.random_world_rotation({"points": G["points"].copy(), "gt_boxes": G["gt_boxes"].copy(), "bm_points": G["bm_in"].copy()},
                                 {"WORLD_ROT_ANGLE": [-0.78539816, 0.78539816], "SAVE_PRE_ROT": True})
    np.testing.assert_array_equal(d["points"], G["wr_points"])
    np.testing.assert_array_equal(d["pre_rot_points"], G["wr_pre_rot_points"])
    np.testing.assert_array_equal(d["pre_rot_points"], G["points"])
    assert float(d["rot_z"]) == float(G["wr_rot_z"]) and abs(float(d["rot_z"])) <= 45.0
    np.testing.assert_array_equal(d["gt_boxes"], G["wr_gt_boxes"])
    np.testing.assert_array_equal(d["bm_points"], G["wr_bm"])


def test_scaling_and_flip():
    np.random.seed(13)
    b, p, sp = ds.global_scaling(G["gt_boxes"].copy(), G["points"].copy(), [0.95, 1.05], [G["bm_in"].copy()])
    for a, ref in ((b, "sc_boxes"), (p, "sc_points"), (sp[0], "sc_bm")):
        np.testing.assert_array_equal(a, G[ref])
    b, p, sp = ds.random_flip_along_x(G["gt_boxes"].copy(), G["points"].copy(), [G["bm_in"].copy()], enable=True)
    for a, ref in ((b, "fl_boxes"), (p, "fl_points"), (sp[0], "fl_bm")):
        np.testing.assert_array_equal(a, G[ref])


def test_best_match_points_from_template_files(tmp_path):
    root = tmp_path / "bm_car"
    root.mkdir()
    frame, ids, is_car = 123, G["bmq_box_ids"], G["bmq_is_car"]
    names = np.array(["Car" if c else "Pedestrian" for c in is_car])
    for i, bid in enumerate(ids):
        if is_car[i]:
            with open(root / ("%d_%d.pkl" % (frame, bid)), "wb") as f:
                pickle.dump(G["bmq_template_%d" % i].reshape(-1), f)
    out = ds.best_match_points(G["gt_boxes"].copy(), names, ids, "%06d" % frame, {"Car": root}, ["Car"])
    np.testing.assert_array_equal(out, G["bmq_points"])
    assert ds.best_match_points(G["gt_boxes"][:0], names[:0], ids[:0], "1", {"Car": root}, ["Car"]).shape == (0, 3)


def test_kitti_bin_round_trip(tmp_path):
    path = tmp_path / "000123.bin"
    ds.write_kitti_bin(path, G["points"])
    assert open(path, "rb").read() == G["bin_bytes"].tobytes()      # the bytes the reference's reader was given
    np.testing.assert_array_equal(ds.read_kitti_bin(path), G["points"])


@pytest.mark.gpu
def test_rotate_scenes_on_device_matches_host_rotation():
    pts = np.concatenate([G["points"], G["points"][:250] * 0.5]).astype(np.float32)
    offs = np.array([0, 600, 850], np.int32)
    rot = np.array([31.5, -12.25], np.float32)
    got = ds.rotate_scenes_on_device(torch.from_numpy(pts).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(rot).cuda()).cpu().numpy()
    ref = np.concatenate([ds.rotate_points_along_z(pts[None, offs[i]:offs[i + 1]], np.array([rot[i] * np.pi / 180.0]))[0] for i in range(2)])
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)          # device cos / sin differ from the host's by an ulp
    np.testing.assert_array_equal(got[:, 2:], pts[:, 2:])
