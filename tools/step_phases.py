"""Per-step anatomy of the pipelined schedule: for every timed step the GPU time (HIP events at the end of the step on the stream its
last kernel runs on), the wall time of the step call, the host seconds of each phase of the three host threads (BTC_TRAINER_TIMING's
accumulators, differenced per step) and the rows of the levels -- to see WHAT differs between the fast and the slow stretches of a run
(the same seeded batches give the same stretches in different processes: profiles/r06_step_phases*.txt).

Also: the CPU every busy thread of the process last ran on and the size of its affinity mask, every 10 steps (that sampling costs a
step ~1.5 ms: every tenth line is slow).

usage: python tools/step_phases.py [steps=120] [warm=64] [features=fp32|bf16]"""
import os
import sys
import time

os.environ["BTC_TRAINER_TIMING"] = "1"
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

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
n_warm = int(sys.argv[2]) if len(sys.argv) > 2 else 64
features = sys.argv[3] if len(sys.argv) > 3 else "fp32"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
pin_to_gpu(0, 0, 1)
torch.manual_seed(666)
np.random.seed(666)
cfg = load_cfg()
if features == "bf16":
    cfg.MODEL.OCC.BACKBONE_3D["FEATURE_DTYPE"] = "bf16"
    cfg.MODEL.BACKBONE_3D["FEATURE_DTYPE"] = "bf16"
model = BtcHotPath(cfg, device=dev).to(dev).train()
use_dist = os.environ.get("BTC_BENCH_FORCE_DIST") == "1"      # the reducer's share: a process group of one rank over RCCL
if use_dist:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    dist.barrier()
tr = HotPathTrainer(model, det_loss=model.det_loss, distributed=use_dist)
step = tr._step
batches = bench.build_batches(n_warm + n_steps + 2, 0, dev)
nb = len(batches)
for i in range(n_warm):
    step(batches[i], batches[i + 1], batches[i + 2])
torch.cuda.synchronize()
marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]
marks[0].record(step.end_stream)
snaps, walls, rows = [dict(step.timing)], [], []


def tasks():
    """(tid -> (name, cpu it last ran on, clock ticks of CPU time)) for every thread of this process"""
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % tid).read()
        except OSError:
            continue
        name = f[f.index("(") + 1:f.rindex(")")]
        rest = f[f.rindex(")") + 2:].split()
        out[int(tid)] = (name, int(rest[36]), int(rest[11]) + int(rest[12]), len(os.sched_getaffinity(int(tid))))
    return out


task_log = []
EVERY = 10
for i in range(n_steps):
    j = n_warm + i
    t0 = time.perf_counter()
    step(batches[j], batches[(j + 1) % nb], batches[(j + 2) % nb])
    walls.append(time.perf_counter() - t0)
    marks[i + 1].record(step.end_stream)
    snaps.append(dict(step.timing))
    rows.append(dict(getattr(model, "last_level_rows", None) or {}))
    if i % EVERY == 0:
        task_log.append((i, tasks()))
torch.cuda.synchronize()
gpu = [marks[i].elapsed_time(marks[i + 1]) for i in range(n_steps)]
keys = ["head", "det_forward", "wait_occ_backward", "det_backward", "det_optimizer", "wait_worker", "cpu_main", "w_occ_backward", "w_opt_prepare", "w_occ_forward",
        "cpu_worker", "p_prepare", "cpu_prep"]
rk = sorted({k for r in rows for k in r})
print("step gpu_ms wall_ms | " + " ".join(keys) + " | " + " ".join(rk))
for i in range(n_steps):
    d = {k: 1e3 * (snaps[i + 1].get(k, 0.0) - snaps[i].get(k, 0.0)) for k in keys}
    print("%3d %.2f %.2f | " % (i, gpu[i], 1e3 * walls[i]) + " ".join("%.2f" % d[k] for k in keys) + " | " + " ".join(str(rows[i].get(k)) for k in rk))
g = np.array(gpu)
print("mean %.3f median %.3f p10 %.3f p90 %.3f -> %.1f scenes/s" % (g.mean(), np.median(g), np.percentile(g, 10), np.percentile(g, 90), 2000.0 / g.mean()))
# correlation of the step time with every column
cols = {k: np.array([1e3 * (snaps[i + 1].get(k, 0.0) - snaps[i].get(k, 0.0)) for i in range(n_steps)]) for k in keys}
for k in rk:
    try:
        cols["rows:" + k] = np.array([float(r.get(k) or 0) for r in rows])
    except (TypeError, ValueError):
        pass
print("correlation with gpu_ms: " + ", ".join("%s %.2f" % (k, np.corrcoef(g, v)[0, 1]) for k, v in cols.items() if v.std() > 0))
# which CPU every busy thread sat on, every EVERY steps (a thread that used > 2 ticks of CPU time since the previous sample)
sib = {}
for c in sorted(os.sched_getaffinity(0)):
    try:
        sib[c] = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
    except OSError:
        sib[c] = "?"
print("affinity:", sorted(os.sched_getaffinity(0)), "siblings:", sib)
print("main tid", os.getpid())
for (i0, a), (i1, b) in zip(task_log[:-1], task_log[1:]):
    busy = [(b[t][2] - a[t][2], t, b[t][0], b[t][1], b[t][3]) for t in b if t in a and b[t][2] - a[t][2] >= 1]
    print("steps %3d-%3d: " % (i0, i1) + "  ".join("%s/%d cpu%d(mask %d) +%d" % (n, t, c, m, d) for d, t, n, c, m in sorted(busy, reverse=True)))
tr.finish()
