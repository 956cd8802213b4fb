"""exclusive host time of the pieces of one sparse layer's forward (tiny scenes; perf_counter wrappers)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
from btcdet_amd.spconv import ops, fused_bn, conv, modules
dev = torch.device("cuda:0")
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
batches = bench.build_batches(2, 0, dev, az_step=4.0)
proc = model.dataset.data_processor
T, C = {}, {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T[label] = T.get(label, 0.0) + time.perf_counter() - t
            C[label] = C.get(label, 0) + 1
    setattr(obj, name, w)
wrap(ops, "_conv_forward", "conv_forward(body)")
wrap(fused_bn, "bn_forward", "bn_forward(body)")
wrap(ops, "build_rulebook_g", "build_rulebook_g")
wrap(ops.SparseConvBNReLUFunction, "forward", "ConvBN.forward(py body)")
wrap(modules.SparseSequential, "forward", "SparseSequential.forward(total, nested)")
wrap(ops, "indice_conv_bn_relu", "indice_conv_bn_relu(total)")
wrap(conv.SparseConvolution, "forward", "SparseConvolution.forward(total)")
def fwd(batch):
    bd = proc.forward_batch(batch["points"], batch["pre_rot_points"], batch["scene_offsets"], batch["rot_z"])
    bd.update({"batch_size": batch["batch_size"], "points": batch["points5"], "gt_boxes": batch["gt_boxes"], "gt_boxes_num": batch["gt_boxes_num"],
               "box_mirr_flag": batch["box_mirr_flag"], "bm_points": batch["bm_points"], "rot_z": batch["rot_z"], "is_train": True})
    t = time.perf_counter()
    ret, tb, _ = model(bd)
    T["model.forward(total)"] = T.get("model.forward(total)", 0.0) + time.perf_counter() - t
for i in range(5):
    fwd(batches[i % 2])
T.clear(); C.clear()
N = 30
for i in range(N):
    fwd(batches[i % 2])
torch.cuda.synchronize()
for k in sorted(T, key=lambda k: -T[k]):
    print("%-36s %8.1f us/step  %5.1f calls/step  %6.1f us/call" % (k, T[k] / N * 1e6, C.get(k, N) / N, T[k] / max(C.get(k, N), 1) * 1e6))
