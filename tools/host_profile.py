"""cProfile of the TRAINING thread of the pipelined step (the other host threads run beside it unprofiled): where the host time of
forward_det / backward goes.  usage: python tools/host_profile.py [steps=60]"""
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
tr = HotPathTrainer(model, det_loss=model.det_loss)
batches = bench.build_batches(4, 0, dev, 2, "kitti")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for i in range(20):
    tr.step(batches[i % 4], batches[(i + 1) % 4])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    tr.step(batches[i % 4], batches[(i + 1) % 4])
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("cumulative").print_stats(45)
out = s.getvalue()
print("\n".join(l[:170] for l in out.splitlines() if l.strip()))
print("(per step: divide tottime / cumtime by %d)" % n)
