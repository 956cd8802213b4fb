"""Full-size golden vectors from the REAL reference, run here (it cannot travel to the GPU box; the vectors can):

    python tests/golden/gen_golden_full.py            # writes tests/golden/btc_full_{a,b,c}.npz

Differences from gen_golden.py (btc_small.npz, 6 k-point scenes):
  * full-size synthetic KITTI scenes (~28 k points): the detection voxelizer's 16 000-voxel cap bites, > 2048 cells pass
    OCC_THRESH so PassOccVox's top-k really selects, 6 seeds over three batches;
  * edge cases: a scene with 0 boxes and no template points (b), an EMPTY scene first (c) and last (d) in the batch, a
    40-point scene and a batch without a `bm_points` key at all (d);
  * the WHOLE per-scene queue of DataProcessor.forward runs as the reference runs it -- mask_points_and_boxes_outside_range
    on raw points that do leave the range, shuffle_points with the seeded global np.random (the permutation is recorded) --
    then the reference's own DatasetTemplate.collate_batch and load_data_to_gpu;
  * the reference's own VoxelBackBoneDeconv, OccHead3D, VoxelBackBone8xOcc and HeightCompression execute over
    tests/golden/oracle_spconv.py (spconv := the C oracle), with name-keyed weights (common.init_by_name): their outputs
    pin the layer graphs of SURVEY §8 a18 / a19 / a23 / a24 by the reference's code rather than by a restatement.
Large float tensors are stored as digests (shape, sums, strided sample; common.digest), integer tensors in full."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_env  # noqa: E402
import oracle_spconv  # noqa: E402

ref_env.install(oracle_spconv)

import torch  # noqa: E402

import common  # noqa: E402
from btcdet.datasets.dataset import DatasetTemplate  # noqa: E402
from btcdet.datasets.processor.data_processor import DataProcessor  # noqa: E402
from btcdet.models import load_data_to_gpu  # noqa: E402
from btcdet.models.backbones_2d.map_to_bev.height_compression import HeightCompression  # noqa: E402
from btcdet.models.backbones_3d.spconv_backbone import VoxelBackBone8xOcc, VoxelBackBoneDeconv  # noqa: E402
from btcdet.models.backbones_3d.vfe.mean_vfe import MeanVFE  # noqa: E402
from btcdet.models.backbones_3d.vfe.occ_vfe import OccVFE  # noqa: E402
from btcdet.models.occ_pnt.occ_dense_heads.occ_head_3D import OccHead3D  # noqa: E402
from btcdet.models.occ_pnt.occ_training_targets.occ_targets_3d import OccTargets3D  # noqa: E402
from btcdet.models.occ_pnt.pass_occ_vox import PassOccVox  # noqa: E402
from btcdet.utils import coords_utils  # noqa: E402

from btcdet_amd import synth  # noqa: E402

MASKS = ["vcc_mask", "voxelwise_mask", "bm_voxelwise_mask", "occ_voxelwise_mask", "fore_voxelwise_mask", "pos_mask",
         "general_cls_loss_mask", "occ_fore_cls_mask", "occ_mirr_cls_mask", "occ_bm_cls_mask", "general_reg_loss_mask"]


def voxel_centers(data_cfg, occ_grid_size):
    """detector3d_template.py:52-63"""
    nx, ny, nz = occ_grid_size
    rng = data_cfg.OCC.POINT_CLOUD_RANGE
    c = coords_utils.get_all_voxel_centers_zyx(1, torch.tensor([nx, ny, nz], dtype=torch.int32), [rng[0], rng[1], rng[2]],
                                               torch.tensor(data_cfg.OCC.VOXEL_SIZE, dtype=torch.float32))[0, ...]
    c = coords_utils.uvd2absxyz(c[2, ...], c[1, ...], c[0, ...], data_cfg.OCC.COORD_TYPE, dim=-1)
    return {"all_voxel_centers": c, "all_voxel_centers_2d": torch.mean(c[:, :, :, :2], dim=0).view(-1, 2)}


def sparse_f32(a):
    a = np.asarray(a)
    nz = np.flatnonzero(a)
    return {"shape": np.array(a.shape), "idx": nz.astype(np.int64), "val": a.reshape(-1)[nz].astype(np.float32)}


def run_case(tag, specs, with_bm_key=True, nets=False):
    cfg = ref_env.load_ref_cfg()
    data_cfg, model_cfg = cfg.DATA_CONFIG, cfg.MODEL
    det_range = np.array(data_cfg.POINT_CLOUD_RANGE, dtype=np.float32)
    occ_range = np.array(data_cfg.OCC.POINT_CLOUD_RANGE, dtype=np.float32)
    proc = DataProcessor(data_cfg.DATA_PROCESSOR, point_cloud_range=occ_range, training=True, occ_config=data_cfg.OCC,
                         det_point_cloud_range=det_range)
    gold, notes, scene_dicts = {}, [], []
    for i, spec in enumerate(specs):
        s, raw, raw_pre, bm = common.raw_scene(spec)
        gold["raw%d_points_sha1" % i] = common.sha1(raw)
        gold["raw%d_n" % i] = np.array(raw.shape[0])
        d = {"points": raw.copy(), "pre_rot_points": raw_pre.copy(), "gt_boxes": s["gt_boxes"].copy(), "rot_z": s["rot_z"],
             "use_lead_xyz": True, "box_mirr_flag": s["box_mirr_flag"]}
        if with_bm_key:
            d["bm_points"] = bm
        perms = []
        real_perm = np.random.permutation

        def recording_perm(n):
            p = real_perm(n)
            perms.append(np.asarray(p).copy())
            return p
        np.random.seed(1000 + spec["seed"])          # tools/train.py seeds the global RNG; shuffle_points draws from it
        np.random.permutation = recording_perm
        try:
            d = proc.forward(d)
        finally:
            np.random.permutation = real_perm
        assert len(perms) == 1
        gold["proc%d_shuffle_idx" % i] = perms[0].astype(np.int32)
        gold["proc%d_gt_boxes" % i] = np.asarray(d["gt_boxes"], np.float32)
        gold["proc%d_points_sha1" % i] = common.sha1(np.asarray(d["points"]))     # = raw[range mask][shuffle_idx]
        gold["proc%d_n_points" % i] = np.array(d["points"].shape[0])
        for k in ["voxel_coords", "voxel_num_points", "det_voxel_coords", "det_voxel_num_points"]:
            gold["proc%d_%s" % (i, k)] = np.asarray(d[k])
        for k in ["voxels", "det_voxels"]:
            gold["proc%d_%s_sha1" % (i, k)] = common.sha1(np.asarray(d[k]))
            gold["proc%d_%s_shape" % (i, k)] = np.array(np.asarray(d[k]).shape)
        d["is_train"] = True
        scene_dicts.append(d)
        print("  scene %d: raw %d -> %d points, %d occ voxels, %d det voxels, %d boxes, %d bm" % (
            i, raw.shape[0], d["points"].shape[0], d["voxel_coords"].shape[0], d["det_voxel_coords"].shape[0], len(d["gt_boxes"]), bm.shape[0]))
    # ---- the reference's collate + upload
    batch = DatasetTemplate.collate_batch(scene_dicts)
    gold["col_points_sha1"] = common.sha1(np.asarray(batch["points"]))
    gold["col_points_shape"] = np.array(batch["points"].shape)
    for k in ["voxel_coords", "det_voxel_coords", "gt_boxes", "box_mirr_flag", "rot_z", "batch_voxel_num", "batch_det_voxel_num"] + (["bm_points"] if with_bm_key else []):
        gold["col_" + k] = np.asarray(batch[k])
    gold["col_gt_boxes_num"] = np.array(batch["gt_boxes_num"])
    gold["col_keys"] = np.array(sorted(batch.keys()))
    bd = dict(batch)
    load_data_to_gpu(bd)
    gold["gpu_dtypes"] = np.array(sorted("%s:%s" % (k, str(v.dtype).replace("torch.", "")) for k, v in bd.items() if torch.is_tensor(v)))
    bd["use_occ_prob"] = np.array([True] * bd["batch_size"])
    grid = proc.occ_grid_size
    vc = voxel_centers(data_cfg, grid)
    B = bd["batch_size"]
    shape = (B, int(grid[2]), int(grid[1]), int(grid[0]))

    targets = OccTargets3D(model_cfg=model_cfg.OCC, voxel_size=proc.occ_voxel_size, point_cloud_range=occ_range,
                           data_cfg=data_cfg, grid_size=grid, num_class=1, voxel_centers=vc)
    bd = targets(bd)
    for k in MASKS:
        gold["tgt_" + k] = np.packbits(bd[k].numpy().astype(bool).reshape(-1))
    gold["tgt_forebox_label"] = np.packbits((bd["forebox_label"].numpy() > 0).reshape(-1))
    gold["tgt_pos_all_num"] = np.array(int(bd["pos_all_num"]))
    for k in ["general_cls_loss_mask_float", "general_reg_loss_mask_float", "res_mtrx"]:
        for kk, vv in sparse_f32(bd[k].numpy()).items():
            gold["tgt_%s_%s" % (k, kk)] = vv
    common.put_digest(gold, "tgt_voxels_absxyz", bd["voxels"].numpy(), n=20000)
    gold["tgt_final_point_mask"] = np.packbits(bd["final_point_mask"].numpy().reshape(-1))
    gold["tgt_final_point_mask_shape"] = np.array(bd["final_point_mask"].shape)

    vfe = MeanVFE(model_cfg=model_cfg.OCC.VFE, num_point_features=4, data_cfg=data_cfg, maxprob=False)
    bd = vfe(bd)
    gold["meanvfe_voxel_features"] = bd["voxel_features"].numpy()

    if nets:
        # ---- the reference's occupancy backbone + head over the oracle-backed spconv (train-mode BatchNorm)
        obb = VoxelBackBoneDeconv(model_cfg.OCC.BACKBONE_3D, input_channels=4, grid_size=np.array(grid))
        head = OccHead3D(model_cfg=model_cfg.OCC, data_cfg=data_cfg, input_channels=obb.num_point_features, num_class=1, grid_size=grid)
        common.init_by_name(obb)
        common.init_by_name(head)
        obb.train()
        head.train()
        with torch.no_grad():
            nd = head(obb(dict(bd)))
        x = nd["encoded_spconv_tensor"]
        gold["net_occ_out_indices_sha1"] = common.sha1(x.indices.numpy().astype(np.int32))
        gold["net_occ_out_n"] = np.array(x.features.shape[0])
        common.put_digest(gold, "net_occ_features", x.features.numpy())
        common.put_digest(gold, "net_pred_occ_logit", nd["pred_occ_logit"].numpy(), n=30000)
        common.put_digest(gold, "net_pred_sem_residuals", nd["pred_sem_residuals"].numpy(), n=30000)
        common.put_digest(gold, "net_batch_pred_occ_prob", nd["batch_pred_occ_prob"].numpy(), n=30000)
        loss, tb = head.get_loss(nd)
        gold["net_head_loss"] = np.array([float(loss), tb["occ_loss_cls"], tb["occ_loss_res"]], dtype=np.float64)
        gold["net_n_candidates"] = np.array([int((nd["batch_pred_occ_prob"][b] > model_cfg.OCC.PARAMS.OCC_THRESH).sum()) for b in range(B)])

    # ---- synthetic head outputs (exactly regenerable: tests/golden/common.py) -> loss, PassOccVox, OccVFE
    logit, res = common.synthetic_head_outputs(B, shape[1], shape[2], shape[3])
    bd["pred_occ_logit"] = torch.from_numpy(logit)
    bd["batch_pred_occ_prob"] = torch.softmax(bd["pred_occ_logit"], dim=1)[:, 1] * bd["general_cls_loss_mask"]
    bd["pred_sem_residuals"] = torch.from_numpy(res)
    gold["pov_n_candidates"] = np.array([int((bd["batch_pred_occ_prob"][b] > model_cfg.OCC.PARAMS.OCC_THRESH).sum()) for b in range(B)])

    class _LossHead(OccHead3D.__mro__[1]):       # OccHeadTemplate
        def __init__(self, *a, **k):
            self.is_softmax = True
            super().__init__(*a, **k)
    lh = _LossHead(model_cfg=model_cfg.OCC, data_cfg=data_cfg, num_class=1, grid_size=grid)
    loss, tb = lh.get_loss(bd)
    gold["head_loss"] = np.array([float(loss), tb["occ_loss_cls"], tb["occ_loss_res"]], dtype=np.float64)

    pov = PassOccVox(model_cfg=model_cfg.OCC, data_cfg=data_cfg, point_cloud_range=det_range, occ_voxel_size=proc.occ_voxel_size,
                     occ_grid_size=proc.occ_grid_size, det_voxel_size=proc.det_voxel_size, det_grid_size=proc.det_grid_size,
                     mode="train", voxel_centers=vc)
    bd = pov(bd)
    gold["pov_voxel_coords"] = bd["voxel_coords"].numpy().astype(np.int32)
    gold["pov_voxel_num_points"] = bd["voxel_num_points"].numpy().astype(np.int32)
    gold["pov_occ_pnts"] = bd["occ_pnts"].numpy()
    gold["pov_added_occ_b_ind"] = bd["added_occ_b_ind"].numpy().astype(np.int32)
    gold["pov_voxels_shape"] = np.array(bd["voxels"].shape)
    v = bd["voxels"].numpy()
    # per-voxel, slot-order independent fingerprint of the merged voxels (the reference's slot order is an unstable sort):
    # float64 sum over the valid slots of every channel
    gold["pov_voxel_slot_sums"] = v.astype(np.float64).sum(1).astype(np.float32)
    ovfe = OccVFE(model_cfg=model_cfg.VFE, num_point_features=6, data_cfg=data_cfg, maxprob=True)
    bd = ovfe(bd)
    gold["occvfe_voxel_features"] = bd["voxel_features"].numpy()
    gold["occvfe_occ_voxel_features"] = bd["occ_voxel_features"].numpy()

    if nets:
        # ---- the reference's detection backbone + HeightCompression over the oracle-backed spconv
        dbb = VoxelBackBone8xOcc(model_cfg.BACKBONE_3D, input_channels=6, grid_size=np.array(proc.det_grid_size),
                                 original_num_rawpoint_features=4)
        common.init_by_name(dbb)
        hc = HeightCompression(model_cfg.MAP_TO_BEV)
        for mode in ("train", "eval"):
            dbb.train(mode == "train")
            state = {k: v.clone() for k, v in dbb.state_dict().items()}
            with torch.no_grad():
                nd = hc(dbb(dict(bd)))
            dbb.load_state_dict(state)
            out, xc = nd["encoded_spconv_tensor"], nd["multi_scale_3d_features"]["x_combine"]
            p = "net_det_%s_" % mode
            gold[p + "out_indices_sha1"] = common.sha1(out.indices.numpy().astype(np.int32))
            gold[p + "xc_indices_sha1"] = common.sha1(xc.indices.numpy().astype(np.int32))
            gold[p + "n"] = np.array([out.features.shape[0], xc.features.shape[0]])
            common.put_digest(gold, p + "out", out.features.numpy())
            common.put_digest(gold, p + "x_combine", xc.features.numpy())
            common.put_digest(gold, p + "spatial_features", nd["spatial_features"].numpy(), n=60000)

    gold["meta_specs"] = np.array([repr(sorted(s.items())) for s in specs])
    gold["meta_with_bm_key"] = np.array(with_bm_key)
    gold["meta_nets"] = np.array(nets)
    gold["meta_notes"] = np.array(notes or ["-"])
    gold["meta_grids"] = np.array([list(proc.occ_grid_size), list(proc.det_grid_size)])
    out = os.path.join(HERE, "btc_%s.npz" % tag)
    np.savez_compressed(out, **gold)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024))
    return gold


CASES = {
    # tag: (scene specs, batch has a bm_points key, run the reference's networks over the oracle-backed spconv)
    "full_a": ([dict(seed=21), dict(seed=22)], True, True),
    "full_b": ([dict(seed=23), dict(seed=24, n_boxes=0, no_bm=True)], True, False),
    "full_c": ([dict(seed=25, keep=0), dict(seed=26)], True, False),             # an EMPTY first scene (the reference skips it)
    "full_d": ([dict(seed=27, keep=40), dict(seed=28, keep=0)], False, False),    # 40 points; empty LAST scene; no bm_points key
}

if __name__ == "__main__":
    which = sys.argv[1:] or list(CASES)
    for tag in which:
        print("==", tag)
        run_case(tag, *CASES[tag])
