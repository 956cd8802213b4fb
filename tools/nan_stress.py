"""pipelined HotPathTrainer for N steps, then: is every parameter finite?  (a non-finite gradient anywhere poisons Adam's moments for
good, so one check at the end catches a transient race).  usage: python tools/nan_stress.py [steps=300] [distinct_batches=8]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
from btcdet_amd.trainer import HotPathTrainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
torch.manual_seed(666); np.random.seed(666)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
tr = HotPathTrainer(model, det_loss=model.det_loss)
batches = bench.build_batches(nb, 0, dev, 2, "kitti")
first_bad = None
for i in range(n):
    tr.step(batches[i % nb], batches[(i + 1) % nb])
    if i % 25 == 24:
        torch.cuda.synchronize()
        bad = [k for k, p in model.named_parameters() if not torch.isfinite(p).all()]
        if bad and first_bad is None:
            first_bad = (i, bad[:4])
            break
torch.cuda.synchronize()
print("steps", i + 1, "non-finite parameters:", first_bad)
