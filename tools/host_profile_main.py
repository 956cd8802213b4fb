"""cProfile of the training thread's share of the pipelined step: BtcHotPath.forward_det (the detection branch's forward, the longest
host phase of the step: BTC_TRAINER_TIMING=1 'det_forward').  usage: python tools/host_profile_main.py [steps=60]"""
import cProfile, io, os, pstats, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from btcdet_amd.affinity import pin_to_gpu
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
from btcdet_amd.trainer import HotPathTrainer
dev = torch.device("cuda", 0); torch.cuda.set_device(0); pin_to_gpu(0, 0, 1)
torch.manual_seed(666); np.random.seed(666)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
pr = cProfile.Profile()
on = [False]
orig = model.forward_det
def wrapped(*a, **k):
    if not on[0]:
        return orig(*a, **k)
    pr.enable()
    try:
        return orig(*a, **k)
    finally:
        pr.disable()
model.forward_det = wrapped
tr = HotPathTrainer(model, det_loss=model.det_loss)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
batches = bench.build_batches(24 + n + 1, 0, dev, 2, "kitti")
for i in range(24):
    tr.step(batches[i], batches[i + 1])
torch.cuda.synchronize()
on[0] = True
for i in range(24, 24 + n):
    tr.step(batches[i], batches[i + 1])
on[0] = False
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(32)
    print("==== forward_det by %s (totals over %d steps: divide by %d)" % (key, n, n))
    print("\n".join(l[:170] for l in s.getvalue().splitlines() if l.strip()))
tr.finish()
