"""Straggler term of the data-parallel step, measured on ONE GPU.

At N > 1 every step ends at the gradient all-reduce -- a barrier -- so a step takes as long as the slowest rank's batch.  Scenes
differ (point counts, active voxels per level), hence so do step times.  This tool runs the bench's step (HotPathTrainer, default
schedule) over >= 64 DISTINCT seeded batches, three times (the first two passes bring the caching allocator and the per-size plans to their
steady state), takes the third pass's per-step times (HIP events at the end of every step on the stream its last kernel runs on)
and reports   predicted_eff_world8 = mean(step) / E[max of 8 independent draws]   (bootstrap over the measured distribution).
No collective is involved: this is the efficiency loss from batch-to-batch variance alone.

usage: python tools/straggler.py [n_batches=64] [out=gpurun_out/r03_straggler.json]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "straggler.json")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
from btcdet_amd.affinity import pin_to_gpu  # noqa: E402
pin_to_gpu(0, 0, 1)
torch.manual_seed(666)
np.random.seed(666)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
tr = HotPathTrainer(model, det_loss=model.det_loss)
# n distinct batches: the scene seeds 8 ranks x n/8 steps would draw (bench.rank_seeds)
batches = []
for r in range(8):
    batches += bench.build_batches(n // 8, r, dev, 2, "kitti")
n = len(batches)
step = tr._step


def one_pass(record):
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    marks[0].record(step.end_stream)
    for i in range(n):
        step(batches[i], batches[(i + 1) % n])
        marks[i + 1].record(step.end_stream)
    torch.cuda.synchronize()
    return np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(n)]) if record else None


one_pass(False)
one_pass(False)   # two warm passes: every batch size has been seen twice, the allocator has its blocks
a0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
ms = one_pass(True)
allocs = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - a0
rng = np.random.default_rng(0)
draws = rng.choice(ms, size=(20000, 8), replace=True)
emax8 = float(draws.max(axis=1).mean())
emax = {w: float(rng.choice(ms, size=(20000, w), replace=True).max(axis=1).mean()) for w in (2, 4, 8)}
res = {"n_batches": n, "points_per_batch": [b["n_points"] for b in batches], "step_ms": [round(float(v), 4) for v in ms],
       "mean_ms": float(ms.mean()), "sd_ms": float(ms.std()), "min_ms": float(ms.min()), "max_ms": float(ms.max()),
       "e_max_ms": {str(w): round(v, 4) for w, v in emax.items()},
       "predicted_eff": {str(w): round(float(ms.mean()) / v, 4) for w, v in emax.items()},
       "predicted_eff_world8": round(float(ms.mean()) / emax8, 4), "device_allocs_in_measured_pass": int(allocs),
       "how": "HotPathTrainer (default schedule) over %d distinct seeded batches, third pass; per-step HIP-event intervals; "
              "E[max of w] by bootstrap (20000 draws)" % n}
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "w") as f:
    json.dump(res, f)
print(json.dumps({k: v for k, v in res.items() if k not in ("step_ms", "points_per_batch")}))
