"""Where the occupancy worker's forward goes (the pipelined schedule's longest host chain once every thread has a CPU of its own): wall
and CPU milliseconds per step of every module of BtcHotPath.occ_module_list, of the detection rulebooks' walk ahead and of the loss, measured
in the running schedule by wrapping the calls.   usage: python tools/worker_phases.py [steps=100] [warm=64]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from btcdet_amd.affinity import pin_to_gpu  # noqa: E402
from btcdet_amd.btc_path import BtcHotPath  # noqa: E402
from btcdet_amd.config import load_cfg  # noqa: E402
from btcdet_amd.trainer import HotPathTrainer  # noqa: E402

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_warm = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
pin_to_gpu(0, 0, 1)
torch.manual_seed(666)
np.random.seed(666)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
T = {}
on = [False]


def wrap(obj, attr, label):
    f = getattr(obj, attr)

    def w(*a, **k):
        if not on[0]:
            return f(*a, **k)
        t0, c0 = time.perf_counter(), time.thread_time()
        try:
            return f(*a, **k)
        finally:
            e = T.setdefault(label, [0.0, 0.0, 0])
            e[0] += time.perf_counter() - t0
            e[1] += time.thread_time() - c0
            e[2] += 1
    setattr(obj, attr, w)


for i, m in enumerate(model.occ_module_list):
    wrap(m, "forward", "occ %d %s" % (i, type(m).__name__))
for i, m in enumerate(model.det_module_list):
    wrap(m, "forward", "det %d %s" % (i, type(m).__name__))
wrap(model.det_modules.backbone_3d, "prefetch_geometry", "det rulebooks ahead (worker)")
wrap(model.occ_modules.occ_dense_head, "get_loss", "occ loss")
wrap(model.occ_modules.occ_dense_head, "premerge", "occ head premerge")
wrap(model, "forward_occ", "forward_occ (whole)")
wrap(model, "forward_det", "forward_det (whole)")
wrap(model, "prepare", "prepare (whole)")
tr = HotPathTrainer(model, det_loss=model.det_loss)
step = tr._step
batches = bench.build_batches(n_warm + n_steps + 2, 0, dev)
for i in range(n_warm):
    step(batches[i], batches[i + 1], batches[i + 2])
torch.cuda.synchronize()
on[0] = True
t0 = time.perf_counter()
for i in range(n_warm, n_warm + n_steps):
    step(batches[i], batches[i + 1], batches[i + 2])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%.3f ms per step (%.1f scenes/s)" % (1e3 * dt / n_steps, 2 * n_steps / dt))
for k, (w, c, n) in sorted(T.items()):
    print("%-44s wall %.3f  cpu %.3f ms per step  (%d calls)" % (k, 1e3 * w / n_steps, 1e3 * c / n_steps, n))
tr.finish()
