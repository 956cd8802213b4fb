"""debug helper: run btc_occ_targets on the golden batch on the GPU and dump the masks"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
from golden_batch import golden_batch
from btcdet_amd.config import load_cfg
from test_hip_occupancy import run_gpu, kitti_batch
cfg = load_cfg()
for which in ["golden", "kitti"]:
    bd = golden_batch()[2] if which == "golden" else kitti_batch()
    out, vc = run_gpu(bd, cfg, torch.device("cuda:0"))
    keys = ["occ_voxelwise_mask", "general_cls_loss_mask", "fore_voxelwise_mask", "bm_voxelwise_mask", "forebox_label", "pos_mask",
            "occ_mirr_cls_mask", "occ_bm_cls_mask", "occ_fore_cls_mask", "general_reg_loss_mask"]
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "occ_dump_%s.npz" % which),
                        **{k: np.packbits(out[k].cpu().numpy() > 0) for k in keys},
                        res_mtrx=out["res_mtrx"].cpu().numpy().astype(np.float16), pos_all_num=int(out["pos_all_num"]))
print("dumped")
