"""Does the step time depend on which hardware queue each stream lands on?  HIP deals streams to GPU_MAX_HW_QUEUES hardware queues round
robin in creation order; n dummy streams created before the trainer's shift the deal.  usage: python tools/queue_probe.py n [steps]"""
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

n_dummy = int(sys.argv[1])
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
pin_to_gpu(0, 0, 1)
torch.manual_seed(666)
np.random.seed(666)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
dummies = []
for _ in range(0 if os.environ.get("DUMMIES_AFTER") == "1" else n_dummy):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0     # hipStreamNonBlocking
    dummies.append(s)
after = os.environ.get("DUMMIES_AFTER") == "1"
if after:
    dummies, n_make = [], n_dummy
    n_dummy = 0
tr = HotPathTrainer(model, det_loss=model.det_loss)
if after:
    for _ in range(n_make):
        s = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
        dummies.append(s)
from btcdet_amd.streams import same_queue  # noqa: E402
four = [torch.cuda.current_stream(), tr.det_stream, tr.prefetch_stream, tr._side_stream]
print("queues_distinct", tr.queues_distinct, "pairs sharing:", [(i, j) for i in range(4) for j in range(i + 1, 4) if same_queue(four[i], four[j])],
      "handles", [hex(s.cuda_stream) for s in four])
step = tr._step
batches = bench.build_batches(64 + n_steps + 2, 0, dev)
for i in range(64):
    step(batches[i], batches[i + 1], batches[i + 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(64, 64 + n_steps):
    step(batches[i], batches[i + 1], batches[i + 2])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("dummy streams %d: %.3f ms per step, %.1f scenes/s" % (n_dummy, 1e3 * dt / n_steps, 2 * n_steps / dt))
tr.finish()
