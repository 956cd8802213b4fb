"""allocated / reserved device memory and hipMalloc count over a long run of the bench step (a leak or a drifting cache shows as growth)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
from btcdet_amd.train_step import GroupOptimizer
from btcdet_amd.spconv import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
occ = [p for p in model.occ_modules.parameters() if p.requires_grad]
det = [p for p in model.det_modules.parameters() if p.requires_grad]
opt = GroupOptimizer([dict(params=occ, lr=3e-3, weight_decay=1e-3, grad_norm_clip=10.0), dict(params=det, lr=1e-2, weight_decay=1e-2, grad_norm_clip=10.0)], 100000)
batches = bench.build_batches(4, 0, dev)
ops.set_defer_wgrad_join(True)
prefetch = torch.cuda.Stream(device=dev, priority=-1)
det_stream = torch.cuda.Stream(device=dev)
step = bench.make_step(model, model, model.dataset.data_processor, [opt], None, prefetch, threaded=True, det_stream=det_stream)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 800
for i in range(N):
    loss = step(batches[i % 4], batches[(i + 1) % 4])
    if i in (50, 100, 200, 400, N - 1):
        torch.cuda.synchronize()
        st = torch.cuda.memory_stats(dev)
        print("step %4d  allocated %.1f MB  reserved %.1f MB  hipMalloc calls %d  loss %.5f" % (i, torch.cuda.memory_allocated(dev) / 2**20, torch.cuda.memory_reserved(dev) / 2**20,
                                                                                       st.get("num_device_alloc", 0), float(loss)))
