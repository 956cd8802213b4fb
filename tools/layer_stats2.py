"""Per-launch table of the sparse-conv kernels in one training step: rows, channels, pairs, pairs per row, time (HIP events,
one stream), algorithmic GB/s and TFLOP/s -- which layers the output-stationary tile wastes its MFMAs on."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
from btcdet_amd.spconv import ops
from btcdet_amd.train_step import GroupOptimizer
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
opt = GroupOptimizer([dict(params=[p for p in model.parameters() if p.requires_grad], lr=1e-3)], 1000)
batches = bench.build_batches(2, 0, dev)
step = bench.make_step(model, model, model.dataset.data_processor, [opt])
for i in range(3):
    step(batches[i % 2])
torch.cuda.synchronize()
prof = ops.LaunchProfile()
ops.PROFILE = prof
step(batches[0])
torch.cuda.synchronize()
ops.PROFILE = None
tot = {}
print("%-11s %7s %4s %4s %8s %6s %8s %8s %7s" % ("kernel", "rows", "cred", "cres", "pairs", "p/row", "us", "GB/s", "TF/s"))
for name, ms, nbytes, flops, info in prof.details():
    if name not in ("conv_apply", "conv_wgrad"):
        continue
    print("%-11s %7d %4d %4d %8d %6.2f %8.1f %8.0f %7.2f" % (name, info["rows"], info["cred"], info["cres"], info["pairs"], info["pairs"] / max(info["rows"], 1),
                                                          ms * 1e3, nbytes / ms / 1e6, flops / ms / 1e9))
    k = (name, info["pairs"] / max(info["rows"], 1) < 6.5)
    tot[k] = tot.get(k, 0) + ms
print({("%s %s" % (k[0], "sparse(<6.5 pairs/row)" if k[1] else "dense")): round(v, 3) for k, v in tot.items()})
