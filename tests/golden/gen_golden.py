"""Generates tests/golden/*.npz by importing the REAL reference (Python, /root/reference) in this
container and running its own modules on seeded synthetic batches.  Run once here; the reference
cannot travel to the GPU box, the vectors can.

    python tests/golden/gen_golden.py

Recipe: stub the absent third-party / compiled modules in sys.modules, alias
np.int / np.float, and rewrite device="cuda" -> "cpu" in torch factory functions.  The stub for
``spconv.utils.VoxelGeneratorV2`` is the repo's CPU oracle voxelizer (spconv itself is absent, see
oracle/btc_oracle.c), everything else that runs is the reference's own code:
  * DataProcessor.forward            (btcdet/datasets/processor/data_processor.py)
  * OccTargets3D.forward             (btcdet/models/occ_pnt/occ_training_targets/occ_targets_3d.py)
  * MeanVFE / OccVFE .forward        (btcdet/models/backbones_3d/vfe/)
  * PassOccVox.forward               (btcdet/models/occ_pnt/pass_occ_vox.py)
  * OccHeadTemplate.get_loss         (btcdet/models/occ_pnt/occ_dense_heads/occ_head_template.py)
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import scipy.spatial  # noqa: E402,F401  (must precede the np.int alias)
import torch  # noqa: E402

np.int = int
np.float = float


# ---- device="cuda" -> "cpu"
def _wrap(fn):
    def w(*a, **k):
        if "device" in k and (k["device"] == "cuda" or str(k["device"]).startswith("cuda")):
            k["device"] = "cpu"
        return fn(*a, **k)
    return w


for name in ["zeros", "ones", "tensor", "as_tensor", "arange", "zeros_like", "ones_like", "rand", "randint", "empty", "full"]:
    setattr(torch, name, _wrap(getattr(torch, name)))


torch.Tensor.cuda = lambda self, *a, **k: self  # loss_utils.py:196 calls .cuda() on a constant


# ---- stub modules
class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        elif isinstance(v, list):
            v = [_EasyDict(x) if isinstance(x, dict) else x for x in v]
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


from oracle import oracle as orc  # noqa: E402


class _Dummy(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


_sp = _mod("spconv", SparseModule=torch.nn.Module, SparseSequential=_Dummy, SubMConv3d=_Dummy, SubMConv2d=_Dummy,
           SparseConv3d=_Dummy, SparseConvTranspose3d=_Dummy, SparseInverseConv3d=_Dummy, SparseMaxPool3d=_Dummy,
           SparseConvTensor=object)
_sp.utils = _mod("spconv.utils", VoxelGeneratorV2=orc.VoxelGeneratorV2, VoxelGenerator=orc.VoxelGeneratorV2)
_mod("easydict", EasyDict=_EasyDict)
_mod("skimage")
_mod("skimage.draw", line_aa=None)
_mod("skimage.io")
for n in ["btcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda", "btcdet.ops.iou3d_nms.iou3d_nms_cuda",
          "btcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda", "btcdet.ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda"]:
    _mod(n)

import yaml  # noqa: E402

from btcdet.datasets.processor.data_processor import DataProcessor  # noqa: E402
from btcdet.models.backbones_3d.vfe.mean_vfe import MeanVFE  # noqa: E402
from btcdet.models.backbones_3d.vfe.occ_vfe import OccVFE  # noqa: E402
from btcdet.models.occ_pnt.occ_dense_heads.occ_head_template import OccHeadTemplate  # noqa: E402
from btcdet.models.occ_pnt.occ_training_targets.occ_targets_3d import OccTargets3D  # noqa: E402
from btcdet.models.occ_pnt.pass_occ_vox import PassOccVox  # noqa: E402
from btcdet.utils import common_utils, coords_utils  # noqa: E402

from btcdet_amd import synth  # noqa: E402
sys.path.insert(0, HERE)
import common  # noqa: E402


def load_ref_cfg():
    cfg = yaml.safe_load(open(os.path.join(REF, "tools/cfgs/model_configs/btcdet_kitti_car.yaml")))
    base = yaml.safe_load(open(os.path.join(REF, "tools", cfg["DATA_CONFIG"]["_BASE_CONFIG_"])))
    data = dict(base)
    data.update({k: v for k, v in cfg["DATA_CONFIG"].items() if k != "_BASE_CONFIG_"})  # config.py:51-68 merge
    cfg["DATA_CONFIG"] = data
    return _EasyDict(cfg)


def voxel_centers(data_cfg, occ_grid_size):
    """detector3d_template.py:52-63"""
    nx, ny, nz = occ_grid_size
    rng = data_cfg.OCC.POINT_CLOUD_RANGE
    grids_num = torch.tensor([nx, ny, nz], dtype=torch.int32)
    voxel_size = torch.tensor(data_cfg.OCC.VOXEL_SIZE, dtype=torch.float32)
    c = coords_utils.get_all_voxel_centers_zyx(1, grids_num, [rng[0], rng[1], rng[2]], voxel_size)[0, ...]
    c = coords_utils.uvd2absxyz(c[2, ...], c[1, ...], c[0, ...], data_cfg.OCC.COORD_TYPE, dim=-1)
    c2d = torch.mean(c[:, :, :, :2], dim=0).view(-1, 2)
    return {"all_voxel_centers": c, "all_voxel_centers_2d": c2d}


def sparse_f32(a):
    a = np.asarray(a)
    nz = np.flatnonzero(a)
    return {"shape": np.array(a.shape), "idx": nz.astype(np.int64), "val": a.reshape(-1)[nz].astype(np.float32)}


def collate(scene_dicts):
    """dataset.py:167-223 restated for the keys of the hot path"""
    B = len(scene_dicts)
    out = {"batch_size": B}
    cat = lambda k: np.concatenate([d[k] for d in scene_dicts], axis=0)
    pad = lambda k: np.concatenate([np.pad(d[k], ((0, 0), (1, 0)), mode="constant", constant_values=i) for i, d in enumerate(scene_dicts)], axis=0)
    for k in ["voxels", "voxel_num_points", "det_voxels", "det_voxel_num_points"]:
        out[k] = cat(k)
    for k in ["points", "voxel_coords", "det_voxel_coords", "bm_points"]:
        out[k] = pad(k)
    maxg = max(len(d["gt_boxes"]) for d in scene_dicts)
    gt = np.zeros((B, maxg, 8), np.float32)
    mirr = np.zeros((B, maxg), np.float32)
    for i, d in enumerate(scene_dicts):
        gt[i, :len(d["gt_boxes"])] = d["gt_boxes"]
        mirr[i, :len(d["gt_boxes"])] = 1.0  # dataset.py:160 builds the flag after box filtering; all synthetic boxes mirror
    out["gt_boxes"], out["box_mirr_flag"] = gt, mirr
    out["gt_boxes_num"] = [len(d["gt_boxes"]) for d in scene_dicts]
    out["rot_z"] = np.array([d["rot_z"] for d in scene_dicts])
    return out


def main(tag, seeds, az_step):
    cfg = load_ref_cfg()
    data_cfg, model_cfg = cfg.DATA_CONFIG, cfg.MODEL
    det_range = np.array(data_cfg.POINT_CLOUD_RANGE, dtype=np.float32)
    occ_range = np.array(data_cfg.OCC.POINT_CLOUD_RANGE, dtype=np.float32)
    proc = DataProcessor(data_cfg.DATA_PROCESSOR, point_cloud_range=occ_range, training=True, occ_config=data_cfg.OCC,
                         det_point_cloud_range=det_range)
    gold = {}
    scenes = []
    for i, seed in enumerate(seeds):
        s = synth.make_scene(seed, az_step=az_step)
        d = {"points": s["points"].copy(), "pre_rot_points": s["pre_rot_points"].copy(), "gt_boxes": s["gt_boxes"].copy(),
             "rot_z": s["rot_z"], "use_lead_xyz": True, "bm_points": s["bm_points"], "box_mirr_flag": s["box_mirr_flag"]}
        # the queue's shuffle step draws from np.random: make it the identity so inputs == synth's (already shuffled)
        perm_backup = np.random.permutation
        np.random.permutation = lambda n: np.arange(n)
        d = proc.forward(d)
        np.random.permutation = perm_backup
        gold["proc%d_gt_boxes" % i] = np.asarray(d["gt_boxes"])
        scenes.append(d)
        for k in ["points", "voxels", "voxel_coords", "voxel_num_points", "det_voxels", "det_voxel_coords", "det_voxel_num_points"]:
            gold["proc%d_%s" % (i, k)] = np.asarray(d[k])
    batch = collate(scenes)
    # load_data_to_gpu: every ndarray -> float32 tensor (models/__init__.py:16-22)
    bd = {}
    for k, v in batch.items():
        bd[k] = torch.from_numpy(np.asarray(v)).float() if isinstance(v, np.ndarray) else v
    bd["is_train"] = True
    bd["use_occ_prob"] = np.array([True] * bd["batch_size"])
    grid = proc.occ_grid_size
    vc = voxel_centers(data_cfg, grid)
    gold["all_voxel_centers"] = vc["all_voxel_centers"].numpy()
    gold["all_voxel_centers_2d"] = vc["all_voxel_centers_2d"].numpy()

    targets = OccTargets3D(model_cfg=model_cfg.OCC, voxel_size=proc.occ_voxel_size, point_cloud_range=occ_range,
                           data_cfg=data_cfg, grid_size=grid, num_class=1, voxel_centers=vc)
    bd = targets(bd)
    for k in ["vcc_mask", "voxelwise_mask", "bm_voxelwise_mask", "occ_voxelwise_mask", "fore_voxelwise_mask", "pos_mask",
              "general_cls_loss_mask", "occ_fore_cls_mask", "occ_mirr_cls_mask", "occ_bm_cls_mask", "general_reg_loss_mask"]:
        gold["tgt_" + k] = np.packbits(bd[k].numpy().astype(bool).reshape(-1))
    gold["tgt_forebox_label"] = np.packbits((bd["forebox_label"].numpy() > 0).reshape(-1))
    gold["tgt_pos_all_num"] = np.array(int(bd["pos_all_num"]))
    for k in ["general_cls_loss_mask_float", "general_reg_loss_mask_float", "res_mtrx"]:
        sp = sparse_f32(bd[k].numpy())
        for kk, vv in sp.items():
            gold["tgt_%s_%s" % (k, kk)] = vv
    gold["tgt_voxels_absxyz"] = bd["voxels"].numpy()
    gold["tgt_final_point_mask"] = bd["final_point_mask"].numpy()

    vfe = MeanVFE(model_cfg=model_cfg.OCC.VFE, num_point_features=4, data_cfg=data_cfg, maxprob=False)
    bd = vfe(bd)
    gold["meanvfe_voxel_features"] = bd["voxel_features"].numpy()

    # synthetic head outputs: exact integer-hash pseudo-random fields (tests/golden/common.py), no storage needed
    B = bd["batch_size"]
    nz, ny, nx = int(grid[2]), int(grid[1]), int(grid[0])
    logit, res = common.synthetic_head_outputs(B, nz, ny, nx)
    bd["pred_occ_logit"] = torch.from_numpy(logit)
    bd["batch_pred_occ_prob"] = torch.softmax(bd["pred_occ_logit"], dim=1)[:, 1] * bd["general_cls_loss_mask"]
    bd["pred_sem_residuals"] = torch.from_numpy(res)

    class _Head(OccHeadTemplate):
        def __init__(self, *a, **k):
            self.is_softmax = True
            super().__init__(*a, **k)
    head = _Head(model_cfg=model_cfg.OCC, data_cfg=data_cfg, num_class=1, grid_size=grid)
    loss, tb = head.get_loss(bd)
    gold["head_loss"] = np.array([float(loss), tb["occ_loss_cls"], tb["occ_loss_res"]], dtype=np.float64)

    pov = PassOccVox(model_cfg=model_cfg.OCC, data_cfg=data_cfg, point_cloud_range=det_range, occ_voxel_size=proc.occ_voxel_size,
                     occ_grid_size=proc.occ_grid_size, det_voxel_size=proc.det_voxel_size, det_grid_size=proc.det_grid_size,
                     mode="train", voxel_centers=vc)
    bd = pov(bd)
    gold["pov_voxels"] = bd["voxels"].numpy()
    gold["pov_voxel_coords"] = bd["voxel_coords"].numpy().astype(np.int32)
    gold["pov_voxel_num_points"] = bd["voxel_num_points"].numpy().astype(np.int32)
    gold["pov_occ_pnts"] = bd["occ_pnts"].numpy()
    gold["pov_added_occ_b_ind"] = bd["added_occ_b_ind"].numpy().astype(np.int32)

    ovfe = OccVFE(model_cfg=model_cfg.VFE, num_point_features=6, data_cfg=data_cfg, maxprob=True)
    bd = ovfe(bd)
    gold["occvfe_voxel_features"] = bd["voxel_features"].numpy()
    gold["occvfe_occ_voxel_features"] = bd["occ_voxel_features"].numpy()

    gold["meta_seeds"] = np.array(seeds)
    gold["meta_az_step"] = np.array(az_step)
    gold["meta_grids"] = np.array([list(proc.occ_grid_size), list(proc.det_grid_size)])
    out = os.path.join(HERE, "btc_%s.npz" % tag)
    np.savez_compressed(out, **gold)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024))
    for k in sorted(gold):
        print("  ", k, gold[k].shape, gold[k].dtype)


if __name__ == "__main__":
    main("small", [11, 12], 0.8)
