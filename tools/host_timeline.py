"""who waits for whom in the steady-state step (full-size scenes, no profiler): host time stamps at the phase boundaries of a
step with NO synchronisation, plus one event per boundary.  lag = (time the GPU reaches the boundary) - (time the host
enqueued it): ~0 means the GPU is starved by the host at that point, large means the host runs ahead of a busy GPU."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
from btcdet_amd.spconv import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
occ = [p for p in model.occ_modules.parameters() if p.requires_grad]
det = [p for p in model.det_modules.parameters() if p.requires_grad]
from btcdet_amd.train_step import GroupOptimizer
opt = GroupOptimizer([{"params": occ, "lr": 3e-3, "weight_decay": 0.001, "grad_norm_clip": 10.0}, {"params": det, "lr": 1e-2, "weight_decay": 0.01, "grad_norm_clip": 10.0}], 74240)
ops.set_defer_wgrad_join(os.environ.get("BTC_DEFER_WGRAD", "1") != "0")
batches = bench.build_batches(2, 0, dev)
proc = model.dataset.data_processor
PREFETCH = os.environ.get("BTC_PREFETCH", "2") != "0"
THREADED = os.environ.get("BTC_PREFETCH", "2") == "2"
if THREADED:
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=1)
side = torch.cuda.Stream(priority=-1) if PREFETCH else None
NAMES = ["zero_grad+voxelize" if not PREFETCH else "zero_grad", "occ branch", "det branch", "loss", "backward" + (" (+prepare in a thread)" if THREADED else ""), "optimizer"] + (["prepare next"] if (PREFETCH and not THREADED) else [])
pending = {}


def step(batch, rec, nxt=None):
    marks = []
    def mark():
        if rec is not None:
            e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((time.perf_counter(), e))
    mark()
    opt.zero_grad(set_to_none=True)
    bd = pending.pop(id(batch), None)
    if bd is None:
        bd = model.assemble(batch)
    ready = bd.pop("__ready_event__", None)
    if ready is not None:
        torch.cuda.current_stream().wait_event(ready)
    n_done = bd.pop("__prepared__", 0)
    bd["use_occ_prob"] = [True, True]
    mark()
    head = model.occ_modules.occ_dense_head
    if hasattr(head, "premerge"):
        head.premerge()
    for mod in model.occ_module_list[n_done:]:
        bd = mod(bd)
    mark()
    for mod in model.det_module_list:
        bd = mod(bd)
    mark()
    loss_occ, tb = head.get_loss(bd)
    loss = loss_occ + bench.MeanSquare.apply(bd["spatial_features"], 1e-3) + bench.MeanSquare.apply(bd["multi_scale_3d_features"]["x_combine"].features, 1e-3)
    mark()
    fut = pool.submit(model.prepare, nxt, side) if (THREADED and PREFETCH and nxt is not None) else None
    loss.backward()
    if fut is not None:
        pending.clear()
        pending[id(nxt)] = fut.result()
    mark()
    opt.step()
    mark()
    if PREFETCH and not THREADED and nxt is not None:
        pending.clear()
        pending[id(nxt)] = model.prepare(nxt, stream=side)
        mark()
    model.mark_step_end()
    if rec is not None:
        rec.append(marks)


nb = len(batches)
for i in range(15):
    step(batches[i % nb], None, batches[(i + 1) % nb])
torch.cuda.synchronize()
rec = []
e0 = torch.cuda.Event(enable_timing=True); e0.record(); torch.cuda.synchronize(); h0 = time.perf_counter()
N = 60
for i in range(15, 15 + N):
    step(batches[i % nb], rec, batches[(i + 1) % nb])
torch.cuda.synchronize()
h1 = time.perf_counter()
print("step %.3f ms (with 7 event records per step)" % ((h1 - h0) / N * 1e3))
host = np.array([[m[0] for m in marks] for marks in rec])
gpu = np.array([[h0 + e0.elapsed_time(m[1]) * 1e-3 for m in marks] for marks in rec])
hd, gd, lag = np.diff(host, axis=1) * 1e3, np.diff(gpu, axis=1) * 1e3, (gpu - host) * 1e3
print("%-20s %9s %9s %12s" % ("phase", "host ms", "gpu ms", "lag at end ms"))
for j, n in enumerate(NAMES):
    print("%-20s %9.3f %9.3f %12.3f" % (n, hd[10:, j].mean(), gd[10:, j].mean(), lag[10:, j + 1].mean()))
print("%-20s %9.3f %9.3f" % ("sum", hd[10:].sum(1).mean(), gd[10:].sum(1).mean()))
print("lag at step start %.3f ms" % lag[10:, 0].mean())
