"""per-launch table of the sparse-conv / rulebook launches of one hot-path step (debug / tuning helper)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
from btcdet_amd.spconv import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
opts = [torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3)]
batches = bench.build_batches(2, 0, dev)
step = bench.make_step(model, model, model.dataset.data_processor, opts)
for i in range(3):
    step(batches[i % 2])
torch.cuda.synchronize()
prof = ops.LaunchProfile(); ops.PROFILE = prof
for i in range(4):
    step(batches[i % 2])
torch.cuda.synchronize(); ops.PROFILE = None
rows = prof.details()
n = len(rows) // 4
print("%-11s %8s %3s %5s %5s %9s %9s %8s %8s" % ("kernel", "rows", "K", "cred", "cres", "pairs", "us", "GB/s", "TF/s"))
for j in range(n):
    name, _, nbytes, flops, info = rows[j]
    ms = np.mean([rows[j + q * n][1] for q in range(4)])
    print("%-11s %8d %3d %5s %5s %9d %9.1f %8.1f %8.2f" % (name, info["rows"], info["K"], info.get("cred", "-"), info.get("cres", "-"),
                                                          info["pairs"], ms * 1e3, nbytes / ms / 1e6, flops / ms / 1e9))
