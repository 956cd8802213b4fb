"""Golden vectors of OccTargets3D under the options the configured model leaves off (missing #3): REVERSE_VIS =
VCC / BACK_TRACK (occ_targets_template.py:110-134) and OCC.DROPOUT_RATE > 0 with and without DROPOUT_RMV (:297-328, :342-343,
:391-392), produced by the REAL reference's module on the small golden batch (tests/golden_batch.py).

    python tests/golden/gen_options_golden.py        -> tests/golden/occ_options.npz

The import of gen_golden installs the stubs / device rewrites of SURVEY.md App. C and imports the reference's classes."""
import copy
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import gen_golden as gg  # noqa: E402  (patches torch factories, registers the stub modules, imports the reference)
import torch  # noqa: E402
from golden_batch import golden_batch  # noqa: E402

MASKS = ["occ_voxelwise_mask", "general_cls_loss_mask", "pos_mask", "general_reg_loss_mask", "occ_fore_cls_mask"]
FLOATS = ["general_cls_loss_mask_float", "general_reg_loss_mask_float"]


def run(reverse_vis="NOTHING", dropout=0.0, rmv=False, seed=0):
    cfg = gg.load_ref_cfg()
    data_cfg, model_cfg = cfg.DATA_CONFIG, cfg.MODEL
    model_cfg.OCC.PARAMS["REVERSE_VIS"] = reverse_vis
    data_cfg.OCC["DROPOUT_RATE"] = dropout
    data_cfg.OCC["DROPOUT_RMV"] = rmv
    occ_range = np.array(data_cfg.OCC.POINT_CLOUD_RANGE, dtype=np.float32)
    det_range = np.array(data_cfg.POINT_CLOUD_RANGE, dtype=np.float32)
    proc = gg.DataProcessor(data_cfg.DATA_PROCESSOR, point_cloud_range=occ_range, training=True, occ_config=data_cfg.OCC,
                            det_point_cloud_range=det_range)
    _, _, bd = golden_batch()
    grid = proc.occ_grid_size
    vc = gg.voxel_centers(data_cfg, grid)
    tgt = gg.OccTargets3D(model_cfg=model_cfg.OCC, voxel_size=proc.occ_voxel_size, point_cloud_range=occ_range, data_cfg=data_cfg,
                          grid_size=grid, num_class=1, voxel_centers=vc)
    coords0 = bd["voxel_coords"].clone()
    np.random.seed(seed)
    torch.manual_seed(seed)
    out = tgt(bd)
    res = {}
    for k in MASKS:
        res[k] = np.packbits(out[k].numpy().astype(bool).reshape(-1))
    for k in FLOATS:
        for kk, vv in gg.sparse_f32(out[k].numpy()).items():
            res["%s_%s" % (k, kk)] = vv
    res["pos_all_num"] = np.array(int(out["pos_all_num"]))
    if dropout > 1e-3:
        c = coords0.long()
        dropped = out["voxel_drop_mask"][c[:, 0], c[:, 1], c[:, 2], c[:, 3]] > 0       # every voxel has its own cell: the drawn set, exactly
        res["dropped"] = np.packbits(dropped.numpy())
        res["n_dropped"] = np.array(int(dropped.sum()))
        res["voxels_sha1"] = gg.common.sha1(out["voxels"].numpy())
        res["n_voxels_out"] = np.array(int(out["voxels"].shape[0]))
        res["fore_voxel_drop_mask"] = np.packbits(out["fore_voxel_drop_mask"].numpy().astype(bool).reshape(-1))
    return res


def main():
    gold = {}
    for tag, kw in (("vcc", dict(reverse_vis="VCC")), ("back_track", dict(reverse_vis="BACK_TRACK")),
                    ("drop", dict(dropout=0.3, seed=5)), ("drop_rmv", dict(dropout=0.3, rmv=True, seed=6))):
        for k, v in run(**kw).items():
            gold["%s_%s" % (tag, k)] = v
        print(tag, "done")
    out = os.path.join(HERE, "occ_options.npz")
    np.savez_compressed(out, **gold)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024))


if __name__ == "__main__":
    main()
