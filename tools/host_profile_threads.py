"""cProfile of the two helper threads of the pipelined step: BtcHotPath.prepare (the next batch's front, prep thread) and
BtcHotPath.forward_occ (worker thread), each profiled inside its own thread.  usage: python tools/host_profile_threads.py [steps=60]"""
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
profs = {"prepare": cProfile.Profile(), "forward_occ": cProfile.Profile()}
on = [False]
for name in profs:
    orig = getattr(model, name)
    def wrapped(*a, _o=orig, _p=profs[name], **k):
        if not on[0]:
            return _o(*a, **k)
        _p.enable()
        try:
            return _o(*a, **k)
        finally:
            _p.disable()
    setattr(model, name, wrapped)
tr = HotPathTrainer(model, det_loss=model.det_loss)
batches = bench.build_batches(4, 0, dev, 2, "kitti")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for i in range(20):
    tr.step(batches[i % 4], batches[(i + 1) % 4])
torch.cuda.synchronize()
on[0] = True
for i in range(n):
    tr.step(batches[i % 4], batches[(i + 1) % 4])
on[0] = False
torch.cuda.synchronize()
for name, pr in profs.items():
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(38)
    print("==== %s (per step: divide by %d)" % (name, n))
    print("\n".join(l[:160] for l in s.getvalue().splitlines() if l.strip()))
