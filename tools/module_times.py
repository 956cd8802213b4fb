"""wall time per hot-path module with a device sync after each (shows where launch-bound glue sits)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
opts = [torch.optim.Adam([p for p in model.occ_modules.parameters() if p.requires_grad], lr=1e-3),
        torch.optim.Adam([p for p in model.det_modules.parameters() if p.requires_grad], lr=1e-3)]
batches = bench.build_batches(2, 0, dev)
proc = model.dataset.data_processor
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return time.perf_counter()
N = 12
for it in range(N + 3):
    if it == 3: T.clear()
    batch = batches[it % 2]
    torch.cuda.synchronize(); t = time.perf_counter()
    for o in opts: o.zero_grad(set_to_none=True)
    bd = proc.forward_batch(batch["points"], batch["pre_rot_points"], batch["scene_offsets"], batch["rot_z"]); t = tick("processor(voxelize x2)", t)
    bd.update({"batch_size": 2, "points": batch["points5"], "gt_boxes": batch["gt_boxes"], "gt_boxes_num": batch["gt_boxes_num"],
               "box_mirr_flag": batch["box_mirr_flag"], "bm_points": batch["bm_points"], "rot_z": batch["rot_z"], "is_train": True,
               "use_occ_prob": np.array([True, True])})
    names = ["occ_targets", "MeanVFE", "VoxelBackBoneDeconv", "OccHead3D", "PassOccVox"]
    for n, m in zip(names, model.occ_module_list):
        bd = m(bd); t = tick(n, t)
    for n, m in zip(["OccVFE", "VoxelBackBone8xOcc", "HeightCompression"], model.det_module_list):
        bd = m(bd); t = tick(n, t)
    loss, tb = model.occ_modules.occ_dense_head.get_loss(bd); t = tick("occ loss", t)
    loss = loss + 1e-3 * bd["spatial_features"].pow(2).mean() + 1e-3 * bd["multi_scale_3d_features"]["x_combine"].features.pow(2).mean(); t = tick("stand-in loss", t)
    loss.backward(); t = tick("backward", t)
    for o in opts: o.step()
    t = tick("2x Adam", t)
tot = sum(T.values())
for k, v in T.items(): print("%-26s %7.3f ms/step" % (k, 1e3 * v / N))
print("%-26s %7.3f ms/step (sum, with a sync after every module)" % ("total", 1e3 * tot / N))
