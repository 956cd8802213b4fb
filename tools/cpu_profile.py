"""host-side profile of one hot-path step (cProfile), to see where the CPU time between launches goes"""
import cProfile, os, pstats, sys, io
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
opts = [torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3)]
batches = bench.build_batches(2, 0, dev)
step = bench.make_step(model, model, model.dataset.data_processor, opts)
for i in range(5):
    step(batches[i % 2])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    step(batches[i % 2])
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(45)
print(s.getvalue()[:9000])
