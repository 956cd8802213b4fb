"""Full-size golden vectors of the REAL reference (tests/golden/gen_golden_full.py: ~28 k-point scenes, voxel caps biting,
empty / 0-box / template-less / 40-point scenes, the reference's own mask + shuffle + collate_batch + load_data_to_gpu, and
its own backbone / head modules executed over the oracle-backed spconv) against
  * the CPU oracle (occ_oracle, the C voxelizer): masks bit for bit, float maps exactly -- same bar as test_oracle_golden;
  * this repository's host-side drop-ins: collate_batch, load_data_to_gpu, DataProcessor's range mask and
    box filter (a1);
  * tests/det_chain.py, the restated VoxelBackBone8xOcc graph the GPU parity tests use: pinned here against what the
    reference's OWN VoxelBackBone8xOcc.forward + HeightCompression produced (a23 / a24)."""
import numpy as np
import pytest
import torch

import common
import det_chain
from golden_batch import FULL_TAGS, golden_batch_full
from oracle import occ_oracle, oracle as orc
from test_oracle_golden import MASKS, canon_slots

from btcdet_amd.config import load_cfg

_CACHE = {}


def case(tag):
    if tag not in _CACHE:
        g, scenes, bd, host = golden_batch_full(tag)
        cfg = load_cfg()
        O = occ_oracle.OccOracle(cfg)
        _CACHE[tag] = (g, scenes, bd, host, cfg, O, O.targets(bd))
    return _CACHE[tag]


@pytest.mark.parametrize("tag", FULL_TAGS)
def test_processor_mask_shuffle_and_voxels(tag):
    """golden_batch_full asserts every per-scene product of the reference's DataProcessor.forward (mask, shuffle, both
    voxelizations) against the oracle; here: the sizes the cases are meant to exercise"""
    g, scenes, bd, host, cfg, O, t = case(tag)
    for i, s in enumerate(scenes):
        assert s["raw_points"].shape[0] > s["points"].shape[0] or s["raw_points"].shape[0] == 0     # the range mask had work to do
        if s["points"].shape[0] > 20000:
            assert s["det_voxel_coords"].shape[0] == 16000                                          # MAX_NUMBER_OF_VOXELS bites
            assert not np.array_equal(g["proc%d_shuffle_idx" % i], np.arange(s["points"].shape[0]))
    if tag in ("full_c", "full_d"):
        assert min(s["points"].shape[0] for s in scenes) == 0                                       # an empty scene


@pytest.mark.parametrize("tag", FULL_TAGS)
def test_range_mask_and_box_filter_of_the_host_protocol(tag):
    """a1 on the host protocol: DataProcessor's range-mask step == the reference's (points, pre_rot_points, gt_boxes)"""
    from btcdet_amd import processor
    from btcdet_amd import synth
    g, scenes, bd, host, cfg, O, t = case(tag)
    for i, s in enumerate(scenes):
        keep = processor.mask_points_by_range(s["raw_points"], synth.KITTI_DET_RANGE)
        assert np.array_equal(common.sha1(s["raw_points"][keep][g["proc%d_shuffle_idx" % i]]), g["proc%d_points_sha1" % i])
        spec_boxes = common.raw_scene(dict(__import__("ast").literal_eval(str(g["meta_specs"][i]))))[0]["gt_boxes"]
        kept = spec_boxes[processor.mask_boxes_outside_range_numpy(spec_boxes, synth.KITTI_DET_RANGE, 1)] if len(spec_boxes) else spec_boxes
        np.testing.assert_array_equal(kept, g["proc%d_gt_boxes" % i])


@pytest.mark.parametrize("tag", FULL_TAGS)
def test_collate_batch_and_load_data_to_gpu_vs_reference(tag):
    g, scenes, bd, host, cfg, O, t = case(tag)
    with_bm = bool(g["meta_with_bm_key"])
    ref_keys = set(str(k) for k in g["col_keys"])
    assert set(host.keys()) == ref_keys, set(host.keys()) ^ ref_keys
    assert np.array_equal(common.sha1(host["points"]), g["col_points_sha1"]) and tuple(host["points"].shape) == tuple(g["col_points_shape"])
    for k in ["voxel_coords", "det_voxel_coords", "gt_boxes", "box_mirr_flag", "rot_z", "batch_voxel_num", "batch_det_voxel_num"] + (["bm_points"] if with_bm else []):
        a, b = np.asarray(host[k]), g["col_" + k]
        assert a.dtype == b.dtype and a.shape == b.shape, (k, a.dtype, b.dtype, a.shape, b.shape)
        np.testing.assert_array_equal(a, b, err_msg=k)
    assert list(host["gt_boxes_num"]) == list(g["col_gt_boxes_num"])
    dtypes = sorted("%s:%s" % (k, str(v.dtype).replace("torch.", "")) for k, v in bd.items() if torch.is_tensor(v))
    assert dtypes == [str(x) for x in g["gpu_dtypes"]]           # everything float32, gt_boxes_num stays a list (App. D.9)


@pytest.mark.parametrize("tag", FULL_TAGS)
def test_targets_vs_reference(tag):
    g, scenes, bd, host, cfg, O, t = case(tag)
    B = bd["batch_size"]
    shape = (B, 9, 157, 209)
    for key in MASKS:
        np.testing.assert_array_equal(t[key].numpy().astype(bool), common.unpack_mask(g, "tgt_" + key, shape), err_msg=key)
    np.testing.assert_array_equal(t["forebox_label"].numpy() > 0, common.unpack_mask(g, "tgt_forebox_label", shape))
    assert int(t["pos_all_num"]) == int(g["tgt_pos_all_num"])
    for k in ["general_cls_loss_mask_float", "general_reg_loss_mask_float", "res_mtrx"]:
        np.testing.assert_array_equal(t[k].numpy(), common.unsparse(g, "tgt_" + k), err_msg=k)
    common.check_digest(g, "tgt_voxels_absxyz", t["voxels"].numpy(), rtol=0, atol=0)
    fpm = tuple(int(v) for v in g["tgt_final_point_mask_shape"])
    np.testing.assert_array_equal(t["final_point_mask"].numpy(), common.unpack_mask(g, "tgt_final_point_mask", fpm))


@pytest.mark.parametrize("tag", FULL_TAGS)
def test_vfe_loss_passoccvox_vs_reference(tag):
    g, scenes, bd, host, cfg, O, t = case(tag)
    B = bd["batch_size"]
    bd2 = dict(bd)
    bd2.update({k: v for k, v in t.items() if not k.startswith("_")})
    np.testing.assert_array_equal(occ_oracle.mean_vfe(bd2["voxels"], bd2["voxel_num_points"]).numpy(), g["meanvfe_voxel_features"])
    logit, res = common.synthetic_head_outputs(B, O.nz, O.ny, O.nx)
    bd2["pred_occ_logit"] = torch.from_numpy(logit)
    bd2["batch_pred_occ_prob"] = torch.softmax(bd2["pred_occ_logit"], dim=1)[:, 1] * bd2["general_cls_loss_mask"]
    bd2["pred_sem_residuals"] = torch.from_numpy(res)
    loss, cls, reg = occ_oracle.occ_losses(bd2, cfg.MODEL.OCC.OCC_DENSE_HEAD.LOSS_CONFIG.LOSS_WEIGHTS)
    np.testing.assert_allclose([float(loss), float(cls), float(reg)], g["head_loss"], rtol=1e-6)
    v, n, c, occ_pnts, ob = occ_oracle.pass_occ_vox(bd2, cfg, orc.revoxelize)
    np.testing.assert_array_equal(c, g["pov_voxel_coords"])
    np.testing.assert_array_equal(n, g["pov_voxel_num_points"])
    assert tuple(v.shape) == tuple(g["pov_voxels_shape"])
    np.testing.assert_allclose(v.astype(np.float64).sum(1), g["pov_voxel_slot_sums"], rtol=1e-6, atol=1e-5)    # slot-order independent
    np.testing.assert_array_equal(occ_pnts, g["pov_occ_pnts"])
    np.testing.assert_array_equal(ob, g["pov_added_occ_b_ind"])
    if tag in ("full_a", "full_b"):
        assert (g["pov_n_candidates"] > cfg.MODEL.OCC.PARAMS.MAX_NUM_OCC_PNTS).all()       # the top-k really selects
    f, o = occ_oracle.occ_vfe(torch.from_numpy(v), torch.from_numpy(n))
    np.testing.assert_allclose(f.numpy(), g["occvfe_voxel_features"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(o.numpy(), g["occvfe_occ_voxel_features"])


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_det_chain_restatement_vs_reference_module(mode):
    """tests/det_chain.py (what the GPU tests of the assembled backbone compare against) == the reference's own
    VoxelBackBone8xOcc.forward + HeightCompression.forward run over the oracle-backed spconv, on a full-size batch with
    name-keyed weights: indices identical, features to fp32 accumulation-order tolerance"""
    from btcdet_amd import backbones_3d
    g = common.load("full_a")
    cfg = load_cfg()
    bb = backbones_3d.VoxelBackBone8xOcc(cfg.MODEL.BACKBONE_3D, input_channels=6, grid_size=np.array([1408, 1600, 40]),
                                         original_num_rawpoint_features=4)
    common.init_by_name(bb)
    sd = {k: v.detach().numpy() for k, v in bb.state_dict().items()}
    coords = g["pov_voxel_coords"]
    rb, lv = det_chain.geometry(coords)
    ref = det_chain.forward_np(sd, rb, lv, g["occvfe_voxel_features"], g["occvfe_occ_voxel_features"], 2, mode == "train")
    p = "net_det_%s_" % mode
    assert np.array_equal(common.sha1(ref["out_indices"].astype(np.int32)), g[p + "out_indices_sha1"])
    assert np.array_equal(common.sha1(ref["x_combine_indices"].astype(np.int32)), g[p + "xc_indices_sha1"])
    assert [ref["out"].shape[0], ref["x_combine"].shape[0]] == list(g[p + "n"])
    for key, a in (("out", ref["out"]), ("x_combine", ref["x_combine"]), ("spatial_features", ref["spatial_features"])):
        scale = float(np.abs(g[p + key + "__sample"]).max())
        err, _ = common.check_digest(g, p + key, a, rtol=0, atol=2e-5 * scale, what=key)
        print("det_chain vs reference module [%s] %s: max |diff| %.2e of scale %.2e" % (mode, key, err, scale))
