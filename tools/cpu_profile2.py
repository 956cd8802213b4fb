"""cProfile of the forward pass only, tiny scenes (host-rate probe), cumulative times of this package's functions"""
import cProfile, os, pstats, sys, io
os.environ.setdefault("BTC_BENCH_AZ_STEP", "4.0")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
batches = bench.build_batches(2, 0, dev)
proc = model.dataset.data_processor
def fwd(batch):
    bd = proc.forward_batch(batch["points"], batch["pre_rot_points"], batch["scene_offsets"], batch["rot_z"])
    bd.update({"batch_size": batch["batch_size"], "points": batch["points5"], "gt_boxes": batch["gt_boxes"], "gt_boxes_num": batch["gt_boxes_num"],
               "box_mirr_flag": batch["box_mirr_flag"], "bm_points": batch["bm_points"], "rot_z": batch["rot_z"], "is_train": True})
    ret, tb, _ = model(bd)
    if os.environ.get("PROFILE_LOSS") == "1":
        loss = ret["loss_occ"] + bench.MeanSquare.apply(ret["spatial_features"], 1e-3) + bench.MeanSquare.apply(ret["x_combine"], 1e-3)
        return loss
    return ret
for i in range(5):
    fwd(batches[i % 2])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    fwd(batches[i % 2])
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats(sys.argv[1] if len(sys.argv) > 1 else "tottime")
ps.print_stats(60)
print(s.getvalue()[:12000])
