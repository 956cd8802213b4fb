"""Which torch (aten) ops one training step issues, by count and host time: the launches that are not ours.
    python tools/op_census.py [--features bf16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
from btcdet_amd.train_step import GroupOptimizer
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
occ = [p for p in model.occ_modules.parameters() if p.requires_grad]
det = [p for p in model.det_modules.parameters() if p.requires_grad]
opt = GroupOptimizer([dict(params=occ, lr=3e-3, weight_decay=1e-3, grad_norm_clip=10.0), dict(params=det, lr=1e-2, weight_decay=1e-2, grad_norm_clip=10.0)], 1000)
batches = bench.build_batches(2, 0, dev)
prefetch = torch.cuda.Stream(device=dev, priority=-1)
from btcdet_amd.spconv import ops
ops.set_defer_wgrad_join(True)
step = bench.make_step(model, model, model.dataset.data_processor, [opt], None, prefetch, threaded=True)
for i in range(6):
    step(batches[i % 2], batches[(i + 1) % 2])
torch.cuda.synchronize()
N = 10
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=False, with_stack=False) as prof:
    for i in range(N):
        step(batches[i % 2], batches[(i + 1) % 2])
    torch.cuda.synchronize()
rows = [(e.key, e.count / N, e.self_cpu_time_total / N, e.cpu_time_total / N) for e in prof.key_averages()]
rows.sort(key=lambda r: -r[3])
print("%-46s %8s %12s %12s" % ("op", "per step", "self us/step", "total us/step"))
for k, c, s, t in rows[:45]:
    print("%-46s %8.1f %12.1f %12.1f" % (k[:46], c, s, t))
