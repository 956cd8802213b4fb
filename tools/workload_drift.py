"""How the step's workload moves with the training state (round 4): the occupancy head is trained by the very steps
that are timed, PassOccVox adds the cells it predicts occupied (OCC_THRESH 0.3, at most MAX_NUM_OCC_PNTS = 2048 a scene) and the
detection levels grow or shrink with that.  Runs HotPathTrainer (default schedule) over N DISTINCT batches, one pass, and prints per
block of 16 steps: mean step ms (HIP events) and the level rows (host integers).

usage: python tools/workload_drift.py [n_steps=256] [out=gpurun_out/drift.json]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from btcdet_amd.btc_path import BtcHotPath  # noqa: E402
from btcdet_amd.config import load_cfg  # noqa: E402
from btcdet_amd.trainer import HotPathTrainer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "drift.json")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
from btcdet_amd.affinity import pin_to_gpu  # noqa: E402
pin_to_gpu(0, 0, 1)
torch.manual_seed(666)
np.random.seed(666)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
tr = HotPathTrainer(model, det_loss=model.det_loss)
batches = bench.build_batches(n + 1, 0, dev, 2, "kitti")
step = tr._step
marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
rows = []
marks[0].record(step.end_stream)
for i in range(n):
    step(batches[i], batches[i + 1])
    marks[i + 1].record(step.end_stream)
    rows.append(dict(model.last_level_rows))
torch.cuda.synchronize()
ms = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(n)])
blocks = []
for b in range(0, n, 16):
    sl = slice(b, min(b + 16, n))
    r = rows[sl]
    blocks.append({"steps": [b, min(b + 16, n)], "mean_ms": round(float(ms[sl].mean()), 3),
                   **{k: int(np.mean([x[k] for x in r])) for k in r[0]}})
    print(blocks[-1])
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "w") as f:
    json.dump({"blocks": blocks, "step_ms": [round(float(v), 4) for v in ms]}, f)
tr.finish()
