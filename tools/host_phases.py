"""host-side time per phase of a step on TINY scenes (GPU work negligible => wall time ~ host / launch cost)"""
import os, sys, time
AZ = None if os.environ.get("HOST_PHASES_FULL") == "1" else 4.0
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
occ = [p for p in model.occ_modules.parameters() if p.requires_grad]
det = [p for p in model.det_modules.parameters() if p.requires_grad]
opt = torch.optim.Adam([{"params": occ}, {"params": det}], lr=1e-3, fused=True)
batches = bench.build_batches(2, 0, dev, az_step=AZ)
proc = model.dataset.data_processor
T = {}
def tick(name, t0):
    torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()
def step(batch, rec):
    t = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    bd = proc.forward_batch(batch["points"], batch["pre_rot_points"], batch["scene_offsets"], batch["rot_z"])
    bd.update({"batch_size": batch["batch_size"], "points": batch["points5"], "gt_boxes": batch["gt_boxes"], "gt_boxes_num": batch["gt_boxes_num"],
               "box_mirr_flag": batch["box_mirr_flag"], "bm_points": batch["bm_points"], "rot_z": batch["rot_z"], "is_train": True})
    if rec: t = tick("voxelize", t)
    use = [True, True]; bd["use_occ_prob"] = use
    for i, mod in enumerate(model.occ_module_list):
        bd = mod(bd)
        if rec: t = tick("occ:" + type(mod).__name__, t)
    for mod in model.det_module_list:
        bd = mod(bd)
        if rec: t = tick("det:" + type(mod).__name__, t)
    loss_occ, tb = model.occ_modules.occ_dense_head.get_loss(bd)
    loss = loss_occ + 1e-3 * bd["spatial_features"].pow(2).mean() + 1e-3 * bd["multi_scale_3d_features"]["x_combine"].features.pow(2).mean()
    if rec: t = tick("loss", t)
    loss.backward()
    if rec: t = tick("backward", t)
    opt.step()
    if rec: t = tick("optimizer", t)
for i in range(5):
    step(batches[i % 2], False)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for i in range(N):
    step(batches[i % 2], False)
torch.cuda.synchronize()
print("step %.3f ms (no per-phase syncs)" % ((time.perf_counter() - t0) / N * 1e3))
for i in range(N):
    step(batches[i % 2], True)
for k, v in T.items():
    print("%-32s %7.3f ms" % (k, v / N * 1e3))
print("sum %.3f ms" % (sum(T.values()) / N * 1e3))
