"""Pins the CPU oracle against golden vectors produced by the REAL reference modules
(tests/golden/gen_golden.py): numpy pre-steps, DataProcessor voxel payloads, OccTargets3D,
MeanVFE / OccVFE, occupancy losses, PassOccVox.  Masks must match bit for bit, floats exactly
(same torch CPU ops in the same order) unless a tolerance is stated."""
import numpy as np
import pytest
import torch

import common  # via tests/golden_batch sys.path
from golden_batch import golden_batch
from oracle import occ_oracle, oracle as orc

from btcdet_amd import synth
from btcdet_amd.config import load_cfg


def canon_slots(v, n):
    """sort the valid slots of every voxel lexicographically (slot order is unspecified in the reference)"""
    v = np.array(v, copy=True)
    for i in np.nonzero(np.asarray(n) > 1)[0]:
        k = int(n[i])
        order = np.lexsort(v[i, :k, ::-1].T)
        v[i, :k] = v[i, :k][order]
    return v


@pytest.fixture(scope="module")
def G():
    g, scenes, bd = golden_batch()
    cfg = load_cfg()
    O = occ_oracle.OccOracle(cfg)
    t = O.targets(bd)
    return g, scenes, bd, cfg, O, t


def test_processor_payloads(G):
    """DataProcessor.forward of the reference (data_processor.py:105-190) == oracle voxelizer on the
    numpy cylinder transform, then voxels[...,1] -= rot_z (float32, padded slots too)."""
    g, scenes, bd, cfg, O, t = G
    occ = orc.VoxelGeneratorV2(synth.KITTI_OCC_VOXEL, synth.KITTI_OCC_RANGE, 12, 20000)
    det = orc.VoxelGeneratorV2(synth.KITTI_DET_VOXEL, synth.KITTI_DET_RANGE, 5, 16000)
    for i, s in enumerate(scenes):
        assert orc.mask_points_by_range(s["points"], synth.KITTI_DET_RANGE).all()
        np.testing.assert_array_equal(g["proc%d_points" % i], s["points"])
        r = occ.generate(orc.absxyz_2_cylinxyz_np(s["pre_rot_points"]))
        v = r["voxels"].copy()
        v[..., 1] = v[..., 1] - s["rot_z"]
        np.testing.assert_array_equal(g["proc%d_voxels" % i], v)
        np.testing.assert_array_equal(g["proc%d_voxel_coords" % i], r["coordinates"])
        np.testing.assert_array_equal(g["proc%d_voxel_num_points" % i], r["num_points_per_voxel"])
        r = det.generate(s["points"])
        np.testing.assert_array_equal(g["proc%d_det_voxels" % i], r["voxels"])
        np.testing.assert_array_equal(g["proc%d_det_voxel_coords" % i], r["coordinates"])


def test_voxel_centers(G):
    g, scenes, bd, cfg, O, t = G
    np.testing.assert_array_equal(O.centers.numpy(), g["all_voxel_centers"])
    np.testing.assert_array_equal(O.centers_2d.numpy(), g["all_voxel_centers_2d"])
    assert (O.nx, O.ny, O.nz) == (209, 157, 9) and (O.snx, O.sny, O.snz) == (214, 157, 49)


MASKS = ["vcc_mask", "voxelwise_mask", "bm_voxelwise_mask", "occ_voxelwise_mask", "fore_voxelwise_mask", "pos_mask",
         "general_cls_loss_mask", "occ_fore_cls_mask", "occ_mirr_cls_mask", "occ_bm_cls_mask", "general_reg_loss_mask"]


@pytest.mark.parametrize("key", MASKS)
def test_target_masks(G, key):
    g, scenes, bd, cfg, O, t = G
    shape = (2, 9, 157, 209)
    np.testing.assert_array_equal(t[key].numpy().astype(bool), common.unpack_mask(g, "tgt_" + key, shape))


def test_target_floats_and_labels(G):
    g, scenes, bd, cfg, O, t = G
    shape = (2, 9, 157, 209)
    np.testing.assert_array_equal(t["forebox_label"].numpy() > 0, common.unpack_mask(g, "tgt_forebox_label", shape))
    assert int(t["pos_all_num"]) == int(g["tgt_pos_all_num"])
    for k in ["general_cls_loss_mask_float", "general_reg_loss_mask_float", "res_mtrx"]:
        np.testing.assert_array_equal(t[k].numpy(), common.unsparse(g, "tgt_" + k))
    np.testing.assert_array_equal(t["voxels"].numpy(), g["tgt_voxels_absxyz"])
    np.testing.assert_array_equal(t["final_point_mask"].numpy(), g["tgt_final_point_mask"])


def test_vfe_loss_passoccvox(G):
    g, scenes, bd, cfg, O, t = G
    bd2 = dict(bd)
    bd2.update({k: v for k, v in t.items() if not k.startswith("_")})
    np.testing.assert_array_equal(occ_oracle.mean_vfe(bd2["voxels"], bd2["voxel_num_points"]).numpy(), g["meanvfe_voxel_features"])
    logit, res = common.synthetic_head_outputs(2, O.nz, O.ny, O.nx)
    bd2["pred_occ_logit"] = torch.from_numpy(logit)
    bd2["batch_pred_occ_prob"] = torch.softmax(bd2["pred_occ_logit"], dim=1)[:, 1] * bd2["general_cls_loss_mask"]
    bd2["pred_sem_residuals"] = torch.from_numpy(res)
    loss, cls, reg = occ_oracle.occ_losses(bd2, cfg.MODEL.OCC.OCC_DENSE_HEAD.LOSS_CONFIG.LOSS_WEIGHTS)
    np.testing.assert_allclose([float(loss), float(cls), float(reg)], g["head_loss"], rtol=1e-6)
    v, n, c, occ_pnts, ob = occ_oracle.pass_occ_vox(bd2, cfg, orc.revoxelize)
    np.testing.assert_array_equal(c, g["pov_voxel_coords"])
    np.testing.assert_array_equal(n, g["pov_voxel_num_points"])
    # the reference orders the points of a cell with an UNSTABLE torch.sort (add_occ_template.py:266): slot order
    # inside a voxel is unspecified there, so voxels are compared as per-voxel multisets of points
    np.testing.assert_array_equal(canon_slots(v, n), canon_slots(g["pov_voxels"], n))
    np.testing.assert_array_equal(occ_pnts, g["pov_occ_pnts"])
    np.testing.assert_array_equal(ob, g["pov_added_occ_b_ind"])
    f, o = occ_oracle.occ_vfe(torch.from_numpy(v), torch.from_numpy(n))
    np.testing.assert_allclose(f.numpy(), g["occvfe_voxel_features"], rtol=1e-6, atol=1e-6)  # slot-order dependent sums
    np.testing.assert_array_equal(o.numpy(), g["occvfe_occ_voxel_features"])
