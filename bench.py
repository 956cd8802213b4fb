#!/usr/bin/env python
"""bench.py -- BtcDet hot path on MI355X: scenes/s forward+backward, KITTI-Car configuration, bs=2 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the hot path over one synthetic batch whose raw points already sit in HBM:
  GPU voxelization of both grids (cylinder occupancy grid + Cartesian detection grid)
  -> OccTargets3D -> MeanVFE -> VoxelBackBoneDeconv -> OccHead3D (+ occupancy loss) -> PassOccVox
  -> OccVFE -> VoxelBackBone8xOcc -> HeightCompression (+ L2 stand-ins for the heads behind the hot path; --heads rpn puts
     BaseBEVBackbone + AnchorHeadSingle and the reference's RPN loss there)
  -> backward (gradient all-reduce over RCCL when N > 1: btcdet_amd/grad_sync.py on a communicator of its own)
  -> the reference's optimizer step per parameter group (norm clip, decoupled weight decay, Adam, OneCycle: btcdet_amd/train_step.py).
The step and its schedules live in the package: btcdet_amd/trainer.py (HotPathTrainer; config.schedule in the JSON names what ran).
Default "pipelined": every step prepares the NEXT batch's weight-independent front (both voxelizations, occupancy targets,
occupancy-branch rulebooks) on a side stream; the detection branch (detached from the occupancy branch, PASS_GRAD False) runs on
its own stream -- its forward beside the occupancy branch's backward, and its backward + all-reduce + optimizer step beside the
occupancy bucket's all-reduce, the occupancy group's optimizer step and the NEXT batch's occupancy-branch forward, which a worker
thread launches with the occupancy weights it has just updated.  K timed steps contain K of everything; every forward pass sees the
weights the in-order loop would give it (tests/test_hip_prefetch.py).  The same schedule runs at N = 1 and N > 1.
BTC_SCHEDULE=in_order|split|pipelined overrides.  fp32 throughout (--features bf16: BASELINE.json configs[2]).
Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      the dominant kernel (conv_apply = fused sparse conv fwd/dgrad) timed live with HIP events
  cpu_baseline  the CPU oracle timed on the host cores for a bounded sample of the same workload (N = 1 only)
  config.in_order_scenes_per_s   the same modules driven in order on one stream (what the reference's own loop gets)
  config.with_rpn_heads          the step with BaseBEVBackbone + AnchorHeadSingle behind the BEV map (second, shorter run).
  config.with_all_heads          ... and the ROI head (proposals, targets, ConvHead, rcnn loss) behind that: the reference's whole loss.
"""
import argparse
import json
import os
import sys
import time

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams on one queue run one after
# the other.  The step keeps four streams busy (occupancy chain, detection chain, next batch's front, weight gradients) and RCCL takes
# queues of its own: eight leave HotPathTrainer room to find a queue for each (btcdet_amd/streams.py).  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_MFMA_PEAK_TF = 157.3   # same guide: fp32-input MFMA peak
BF16_MFMA_PEAK_TF = 2500.0  # same guide: dense bf16 MFMA peak (AMD's 5 PF figure is 2:1 sparse)


def pmc_traffic_per_launch(waymo=False, bf=False):
    """HBM bytes per conv_apply launch of this configuration from the committed PMC passes (profiles/r06_pmc*.json, else r05: rocprofv3 --pmc
    FETCH_SIZE and --pmc WRITE_SIZE in separate runs of the same bench command, FETCH_SIZE doubled per the gfx950 note of
    MI355X_MICROARCH.md); PMC counters cannot be read inside the timed process, so this is null when there is no pass of the
    configuration on file"""
    sfx = ("_waymo" if waymo else "") + ("_bf16" if bf else "")
    names = ["r06_pmc%s.json" % sfx, "r05_pmc%s.json" % sfx] + ([] if sfx else ["r04_pmc.json", "r03j_pmc.json", "r03_pmc.json", "r02h_pmc.json"])
    try:
        name = next(n for n in names if os.path.exists(os.path.join(ROOT, "profiles", n)))
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)["conv_apply"]["hbm_bytes_per_launch"]
    except (StopIteration, OSError, KeyError, ValueError):
        return None


def scheduled_conv_ms(waymo=False, bf=False):
    """conv_apply (+ split_reduce) milliseconds per step as the kernels run IN the step, from the committed rocprofv3 kernel statistics of
    the same command: {"serial": in order with the weight gradients on their side stream, "default": the pipelined schedule}.  The
    roofline leg below times the launches alone (nothing beside them); these say what contention costs.  None where no file exists."""
    import csv
    sfx = ("_waymo" if waymo else "") + ("_bf16" if bf else "")
    out = {}
    for kind, name in (("serial", "r06_serial%s_kernel_stats.csv" % sfx), ("default", "r06_bench%s_kernel_stats.csv" % sfx)):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rows = list(csv.DictReader(f))
            steps = sum(int(r["Calls"]) for r in rows if "adam_apply" in r["Name"]) / 2.0
            ns = sum(int(r["TotalDurationNs"]) for r in rows if "conv_apply" in r["Name"] or "split_reduce" in r["Name"])
            if steps > 0:
                out[kind] = {"ms_per_step": round(ns / steps / 1e6, 4), "source": "profiles/" + name}
        except (OSError, KeyError, ValueError):
            pass
    return out or None


def rank_seeds(rank, i, batch_size=2):
    """scene seeds of batch i on a rank: ranks draw disjoint scenes (the DistributedSampler shard)"""
    return [1000 * rank + 10 * i + j for j in range(batch_size)]


def max_over_ranks(dt, dist, device):
    """wall time of the slowest rank (the step ends at the gradient all-reduce barrier)"""
    if dist is None:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def build_batches(n_batches, rank, device, batch_size=2, profile="kitti", az_step=None):
    """az_step: azimuth step of the synthetic scan in degrees (tools: tiny scenes, whose step time is host / launch cost)"""
    from btcdet_amd import synth
    batches = []
    for i in range(n_batches):
        seeds = rank_seeds(rank, i, batch_size)
        kw = {"az_step": float(az_step)} if az_step else {}
        b = synth.make_batch(seeds, profile=profile, **kw)
        batches.append({
            "batch_size": batch_size,
            "points5": torch.from_numpy(b["points"]).to(device),                       # [b,x,y,z,i] (collate layout)
            "points": torch.from_numpy(np.ascontiguousarray(b["points"][:, 1:])).to(device),
            "pre_rot_points": torch.from_numpy(b["pre_rot_points"]).to(device),
            "scene_offsets": torch.from_numpy(b["scene_offsets"]).to(device),
            "gt_boxes": torch.from_numpy(b["gt_boxes"]).to(device),
            "gt_boxes_num": torch.tensor(b["gt_boxes_num"], dtype=torch.int32, device=device),
            "box_mirr_flag": torch.from_numpy(b["box_mirr_flag"]).to(device),
            "bm_points": torch.from_numpy(b["bm_points"]).to(device),
            "rot_z": torch.from_numpy(b["rot_z"]).to(device),
            "n_points": int(b["points"].shape[0]),
        })
    return batches


# the training step, its schedules and the stand-in loss live in the package (btcdet_amd/trainer.py: HotPathTrainer, make_step);
# the names stay importable from here for the tests and tools that grew up with them
from btcdet_amd.trainer import MeanSquare, make_step  # noqa: E402,F401


def cpu_baseline(seconds_budget=25.0, max_scenes=2):
    """CPU oracle ("port") on ONE host core: per scene the reference's CPU dataset/voxelize path (cylinder transform +
    both voxelizations, data_processor.py:105-190) and every sparse-conv layer of both backbones + the occupancy head
    (rulebooks, forward, dgrad, wgrad) with the C oracle.  Not in the sample: occupancy targets, VFEs, losses,
    PassOccVox merging, BatchNorm, dense scatter, optimizer (all small next to the convolutions on a CPU)."""
    from btcdet_amd import synth
    from oracle import oracle as orc
    occ_gen = orc.VoxelGeneratorV2(synth.KITTI_OCC_VOXEL, synth.KITTI_OCC_RANGE, 12, 20000)
    det_gen = orc.VoxelGeneratorV2(synth.KITTI_DET_VOXEL, synth.KITTI_DET_RANGE, 5, 16000)
    rng = np.random.default_rng(0)
    C, S, T = orc.MODE_CONV, orc.MODE_SUBM, orc.MODE_TRANSPOSE
    # (cin, cout, kernel, stride, padding, mode, source tensor, result tensor); layer tables of spconv_backbone.py:106-128,656-767
    occ_layers = [(4, 16, 3, 1, 1, C, "in", "a"), (16, 32, 3, 2, 1, C, "a", "b"), (32, 32, 3, 1, 0, S, "b", "b"),
                  (32, 64, 3, 2, 1, C, "b", "c"), (64, 64, 3, 1, 0, S, "c", "c"), (64, 32, 3, 2, 1, T, "c", "d"),
                  (32, 32, 3, 1, 0, S, "d", "d"), (32, 32, 3, 2, 1, T, "d", "e"), (32, 32, 3, 1, 0, S, "e", "e"),
                  (32, 2, 3, 1, 0, S, "e", "cls"), (32, 3, 3, 1, 0, S, "e", "res")]
    det_layers = [(6, 16, 3, 1, 0, S, "in", "x1"), (16, 16, 3, 1, 0, S, "x1", "x1"), (16, 32, 3, 2, 1, C, "x1", "x2"),
                  (32, 32, 3, 1, 0, S, "x2", "x2"), (32, 32, 3, 1, 0, S, "x2", "x2"), (32, 64, 3, 2, 1, C, "x2", "x3"),
                  (64, 64, 3, 1, 0, S, "x3", "x3"), (64, 64, 3, 1, 0, S, "x3", "x3"), (64, 64, 3, 2, (0, 1, 1), C, "x3", "x4"),
                  (64, 64, 3, 1, 0, S, "x4", "x4"), (64, 64, 3, 1, 0, S, "x4", "x4"), (64, 128, (3, 1, 1), (2, 1, 1), 0, C, "x4", "out"),
                  (32, 32, 3, 2, 1, C, "x2", "d2"), (32, 64, 3, 2, (0, 1, 1), C, "d2", "d2b"), (64, 64, 3, 2, (0, 1, 1), C, "x3", "d3"),
                  (128, 64, (2, 1, 1), (2, 1, 1), 0, C, "out", "bev"), (256, 128, 3, 1, 0, S, "x4", "x4c"), (128, 128, 3, 1, 0, S, "x4c", "x4c")]

    def run_chain(layers, idx0, shape0, c0):
        tensors = {"in": (idx0, list(shape0), (rng.standard_normal((idx0.shape[0], c0))).astype(np.float32))}
        cache = {}
        for cin, cout, k, st, pd, mode, src, dst in layers:
            idx, shape, feat = tensors[src]
            if feat.shape[1] != cin:
                feat = np.ascontiguousarray(np.resize(feat, (feat.shape[0], cin)))
            key = (src, str(k), str(st), str(pd), mode)
            if key not in cache:
                cache[key] = orc.rulebook(idx, shape, k, st, pd, 1, mode)
            o_idx, nbr_out, nbr_in, osh = cache[key]
            W = (rng.standard_normal((27, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)[:nbr_out.shape[1]]
            out = orc.conv_fwd(feat, W, None, nbr_out)
            dout = np.ones_like(out)
            orc.conv_dgrad(dout, W, nbr_in)
            orc.conv_wgrad(feat, dout, nbr_out, W.shape)
            tensors[dst] = (o_idx, list(osh), np.maximum(out, 0))

    t_start = time.perf_counter()
    scenes, t_vox, t_conv = 0, 0.0, 0.0
    while scenes < max_scenes and (scenes == 0 or time.perf_counter() - t_start < seconds_budget / 2):
        s = synth.make_scene(5000 + scenes)
        t0 = time.perf_counter()
        cyl = orc.absxyz_2_cylinxyz_np(s["pre_rot_points"])
        r = occ_gen.generate(cyl)
        r["voxels"][..., 1] -= s["rot_z"]
        rd = det_gen.generate(s["points"])
        t1 = time.perf_counter()
        run_chain(occ_layers, np.pad(r["coordinates"], ((0, 0), (1, 0))).astype(np.int32), [9, 157, 209], 4)
        run_chain(det_layers, np.pad(rd["coordinates"], ((0, 0), (1, 0))).astype(np.int32), [41, 1600, 1408], 6)
        t2 = time.perf_counter()
        t_vox += t1 - t0
        t_conv += t2 - t1
        scenes += 1
    total = t_vox + t_conv
    # the same voxelize path with one worker process per host core (the reference's DataLoader-worker parallelism)
    n_procs = os.cpu_count() or 1
    try:  # each worker holds spconv's dense 40 x 1600 x 1408 int32 scratch grid (360 MB): stay below 1/4 of the RAM
        avail = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) * 1024
        n_procs = max(1, min(n_procs, int(0.25 * avail / 0.5e9)))
    except Exception:
        n_procs = min(n_procs, 16)
    vox_all, vox_procs = None, 0
    try:
        from oracle import cpu_voxel_bench
        rate, vox_procs = cpu_voxel_bench.all_cores(n_procs)
        vox_all = round(rate, 1)
    except Exception as e:  # the baseline is a report, not a gate
        print("all-cores voxelize baseline skipped: %r" % (e,), file=sys.stderr)
    return {"value": round(scenes / total, 3), "unit": "scenes/s", "cores": 1, "kind": "port",
            "sample": "%d synthetic KITTI scene(s), C oracle, 1 thread: cylinder transform + 2 voxelizations (%.1f ms/scene) + all %d sparse-conv "
                      "layers of both backbones and the occupancy head: rulebooks + fwd + dgrad + wgrad (%.0f ms/scene); occupancy targets, "
                      "VFEs, BatchNorm, PassOccVox merge, losses and optimizer are NOT in the sample"
                      % (scenes, 1e3 * t_vox / scenes, len(occ_layers) + len(det_layers), 1e3 * t_conv / scenes),
            "voxelize_scenes_per_s": round(scenes / t_vox, 2), "voxelize_all_cores_scenes_per_s": vox_all, "voxelize_all_cores_procs": vox_procs,
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--priming", type=int, default=64, help="optimizer steps before the timed region, warm-up included (see the comment at `priming`)")
    ap.add_argument("--no-extras", action="store_true", help="skip the two secondary measurements (in-order rate, RPN-head run)")
    ap.add_argument("--workload", choices=["kitti", "waymo"], default="kitti",
                    help="kitti: the configuration BASELINE.json's metric is quoted on (default); waymo: the Waymo-shaped synthetic "
                         "scenes of configs[4] (~166 k points/scene, 1504 x 1504 x 40 grid) -- same path, 6x the work per scene")
    ap.add_argument("--features", choices=["fp32", "bf16"], default="fp32",
                    help="fp32: the reference's precision (default, the headline number); bf16: BASELINE.json configs[2] -- bfloat16 "
                         "activations between sparse layers, fp32 weights / accumulation / statistics")
    ap.add_argument("--heads", choices=["standin", "rpn", "full"], default="standin",
                    help="standin (headline): L2 stand-ins on the two tensors the heads behind the hot path consume; rpn: BaseBEVBackbone + "
                         "AnchorHeadSingle with the reference's RPN loss (btcnet.py:108-114) behind the BEV map, the ROI head's tensor keeps "
                         "its stand-in (ConvHead is not built)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback for the hot path)"
    dev_index = local_rank            # one rank per GPU over RCCL ("nccl" on ROCm)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # host threads of this rank next to its GPU (btcdet_amd/affinity.py): the step is launch-rate sensitive, and on a two-socket
    # host an unpinned process that lands on the far socket loses 3-6 % (and makes the number box-dependent)
    from btcdet_amd.affinity import pin_to_gpu
    all_cpus = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    pinned = pin_to_gpu(dev_index, local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    dist = None

    class _StdoutToStderr(object):
        """RCCL prints a version banner to STDOUT when a communicator is created; this script's stdout is ONE JSON line.  The banner goes
        through C stdio, which buffers fully when stdout is a pipe (the driver's case): without the fflush below it sits in that buffer
        past the redirection and comes out on the real stdout at exit, BEHIND the JSON line"""

        @staticmethod
        def _fflush():
            import ctypes
            try:
                ctypes.CDLL(None).fflush(None)
            except (OSError, AttributeError):
                pass

        def __enter__(self):
            sys.stdout.flush()
            self._fflush()
            self.saved = os.dup(1)
            os.dup2(2, 1)

        def __exit__(self, *a):
            sys.stdout.flush()
            self._fflush()
            os.dup2(self.saved, 1)
            os.close(self.saved)

    # BTC_BENCH_FORCE_DIST=1: take the distributed path (process group, reducer, barriers) at world size 1 too --
    # the way to exercise the RCCL code path on a single-GPU box
    use_dist = world > 1 or os.environ.get("BTC_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with _StdoutToStderr():
            dist.init_process_group(backend="nccl", device_id=device)
            dist.barrier()   # the communicator exists (and has printed its banner) before anything else happens

    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    from btcdet_amd.spconv import ops
    from btcdet_amd.trainer import HotPathTrainer, reference_groups
    from btcdet_amd.train_step import GroupOptimizer

    waymo = args.workload == "waymo"
    bs = 2

    def build_model(heads):
        torch.manual_seed(666)
        np.random.seed(666 + rank)
        cfg = load_cfg(os.path.join(ROOT, "btcdet_amd", "cfgs", "btcdet_waymo_synth.yaml") if waymo else None)
        if args.features == "bf16":
            cfg.MODEL.OCC.BACKBONE_3D["FEATURE_DTYPE"] = "bf16"
            cfg.MODEL.BACKBONE_3D["FEATURE_DTYPE"] = "bf16"
        m = BtcHotPath(cfg, device=device, heads=heads).to(device)
        m.train()
        return m

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    model = build_model(args.heads if args.heads in ("rpn", "full") else None)
    # The workload is FIXED: every step (priming, warm-up, timed) draws a batch of scenes nobody has seen, as an epoch of the reference's
    # loop does (tools/train_utils/train_utils.py:96-124: one new batch per iteration).  Round 3 recycled 4 batches; the live optimizer
    # memorised them, the occupancy head stopped predicting occupied cells, PassOccVox added few voxels and the detection levels shrank
    # while they were being timed.  That line survives as config.recurring_batches_scenes_per_s.
    # priming + warm-up = 64 optimizer steps: the caching allocator and the plans reach their steady state within ~16; the rest lets the
    # occupancy head leave its random initialisation, so PassOccVox adds what a head in training adds (tools/workload_drift.py: the
    # detection levels hover around 32-34 K / 30-36 K / 14-19 K / 6-8 K rows from step ~30 on; config.level_rows reports the timed steps')
    priming = max(0, args.priming - args.warmup)
    n_distinct = min(priming + args.warmup + args.steps + 2 + (0 if (args.no_extras or waymo) else 12) + (0 if args.no_roofline else 8),
                     192 if not waymo else 48)
    batches = build_batches(n_distinct, rank, device, bs, args.workload)
    nb = len(batches)
    cursor = [0]       # next unseen batch (wraps around only past 192 distinct ones; Waymo shape: 48)
    # Schedule: HotPathTrainer's default ("pipelined": detection branch on its own stream, occupancy branch one step ahead, each
    # thread's bucket all-reduced behind its backward when a process group exists).  BTC_SCHEDULE=in_order|split|pipelined
    # overrides.
    schedule = os.environ.get("BTC_SCHEDULE") or "pipelined"
    # the reference's optimizer step per parameter group (tools/train_utils/train_utils.py:121-124; yaml:331-372): gradient-norm
    # clip at 10, adam_onecycle = decoupled weight decay + Adam(betas=(mom, 0.99)) with lr / mom on the OneCycle schedule of a
    # 40-epoch run over KITTI's 3712 training frames -- btcdet_amd/train_step.py (checked against the reference's own
    # OptimWrapper / OneCycle, tests/test_train_step_cpu.py).  The two optimizers are the two groups of one object.
    with_reducer = use_dist
    with _StdoutToStderr():
        trainer = HotPathTrainer(model, schedule=schedule, distributed=with_reducer, det_loss=model.det_loss)
    step, grad_sync, opt = trainer._step, trainer.grad_sync, trainer.optimizer

    def timed_run(step_fn, n_steps, n_warm, end_stream, pool=None, rows_of=None):
        """n_warm untimed steps, then exactly n_steps timed ones bracketed by barrier + synchronize -> (seconds, per-step ms, hipMallocs,
        per-step level rows).  pool: the batches to cycle through (default: the distinct ones, continuing where the last run stopped);
        rows_of: a model whose last_level_rows (host integers, no read-back) are logged per timed step"""
        if pool is None:
            pool, base = batches, cursor[0]
            cursor[0] += n_warm + n_steps
        else:
            base = 0
        n = len(pool)
        two_ahead = getattr(step_fn, "pipelined", False)   # a loader's second prefetched batch: its weight-independent front runs one step earlier
        call = (lambda j: step_fn(pool[(base + j) % n], pool[(base + j + 1) % n], pool[(base + j + 2) % n])) if two_ahead else \
            (lambda j: step_fn(pool[(base + j) % n], pool[(base + j + 1) % n]))
        for i in range(n_warm):
            call(i)
        sync()
        if getattr(step_fn, "timing", None):
            step_fn.timing.clear()     # (BTC_TRAINER_TIMING=1: host phases of the timed steps only)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]
        allocs0 = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
        rows = []
        t0 = time.perf_counter()
        marks[0].record(end_stream)
        for i in range(n_warm, n_warm + n_steps):  # each step prepares its successor: K steps, K preparations
            call(i)
            marks[i - n_warm + 1].record(end_stream)  # end of the step's work on the stream its last kernel runs on (no host wait)
            if rows_of is not None:
                rows.append(getattr(rows_of, "last_level_rows", None))
        sync()
        dt = time.perf_counter() - t0
        ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(n_steps)]
        ms = sorted(ms)
        return dt, ms, torch.cuda.memory_stats(device).get("num_device_alloc", 0) - allocs0, rows

    # Setup, before the W warmup steps: the caching allocator, the per-batch-size geometry plans and the flat optimizer buffers reach
    # their steady state only after every one of the 4 synthetic batches has been seen a few times; a short W (the driver's choice)
    # would otherwise leave hipMalloc calls and plan construction inside the timed region (config.priming_steps in the JSON).
    dt, per_step_ms, device_allocs, level_rows = timed_run(step, args.steps, priming + args.warmup, step.end_stream, rows_of=model)
    dt = max_over_ranks(dt, dist, device)
    if getattr(step, "timing", None) and rank == 0:   # BTC_TRAINER_TIMING=1: host milliseconds per phase and step of the timed region
        n = max(step.timing.get("n", 1), 1)
        print("trainer host ms per step:", {k: round(1e3 * v / n, 3) for k, v in step.timing.items() if k != "n"}, file=sys.stderr)

    # ---- secondary measurements (outside the timed region, N = 1 reporting only; --no-extras skips them)
    extras = {}
    want_extras = not args.no_extras and not waymo
    plain_step = make_step(model, model, model.dataset.data_processor, [opt], grad_sync, det_loss=model.det_loss)   # one stream, one thread
    if want_extras:
        # what the reference's own loop (train_one_epoch_multi_opt) gets from these modules unchanged: in order, one stream
        k_in = min(args.steps, 10)
        dt_in, _, _, _ = timed_run(lambda b, nxt: plain_step(b), k_in, 2, None)
        dt_in = max_over_ranks(dt_in, dist, device)
        extras["in_order_scenes_per_s"] = round(bs * world * k_in / dt_in, 2)
    # roofline leg: the SAME steps once more, in order, with a HIP event pair around every sparse-conv / rulebook launch
    # (kept out of the timed region above because counting the pairs of each rulebook needs a read-back)
    prof = None
    prof_steps = 0
    if not args.no_roofline and rank == 0:
        prof = ops.LaunchProfile()
        ops.PROFILE = prof
        prof_steps = min(args.steps, 8)
    if not args.no_roofline:
        for i in range(min(args.steps, 8)):
            plain_step(batches[(cursor[0] + i) % nb])
        cursor[0] += min(args.steps, 8)
        sync()
    ops.PROFILE = None
    if want_extras and step.pipelined:
        # round 3's headline, kept for continuity: the same schedule over FOUR recurring batches, which the live optimizer memorises
        # (after ~15 steps the occupancy head predicts few occupied cells and the detection levels shrink)
        k_rec = min(args.steps, 20)
        dt_rec, _, _, rows_rec = timed_run(step, k_rec, 16, step.end_stream, pool=batches[:4], rows_of=model)
        dt_rec = max_over_ranks(dt_rec, dist, device)
        extras["recurring_batches_scenes_per_s"] = {"scenes_per_s": round(bs * world * k_rec / dt_rec, 2), "steps": k_rec, "warm": 16,
                                                    "level_rows_last_step": rows_rec[-1] if rows_rec else None,
                                                    "what": "4 recurring batches (round 3's workload): NOT the headline -- the optimizer memorises them"}
    if want_extras and args.heads == "standin":
        # the same step with the §8f row-1 heads behind the BEV map (BaseBEVBackbone + AnchorHeadSingle, RPN loss of btcnet.py:108-114;
        # dense 2-D convs = vendor library): a second model, trainer and optimizer, run after the headline measurement
        del plain_step
        model_r = build_model("rpn")
        with _StdoutToStderr():
            tr_r = HotPathTrainer(model_r, schedule=schedule, distributed=with_reducer, det_loss=model_r.det_loss)
        # 32 untimed steps first: the vendor library's convolution searches, the allocator and the thread placement of a NEW model and trainer
        # (12 were not enough: 88.7 in this line against 164.5 for `bench.py --heads rpn` alone, same box)
        k_r = min(args.steps, 20)
        dt_r, ms_r, _, _ = timed_run(tr_r._step, k_r, 32, tr_r._step.end_stream)
        dt_r = max_over_ranks(dt_r, dist, device)
        extras["with_rpn_heads"] = {"scenes_per_s": round(bs * world * k_r / dt_r, 2), "ms_per_step": round(1e3 * dt_r / k_r, 3), "steps": k_r,
                                    "what": "BaseBEVBackbone + AnchorHeadSingle (RPN cls / loc / dir loss, targets assigned in the prepared front) "
                                            "behind HeightCompression; x_combine keeps its L2 stand-in"}
        # ... and with the ROI head behind the proposals too: get_training_loss = loss_rpn + loss_rcnn + occupancy loss (btcnet.py:58-129)
        tr_r.finish()    # (its host threads and their queued work are gone before the next model runs)
        del tr_r, model_r
        torch.cuda.empty_cache()
        model_f = build_model("full")
        with _StdoutToStderr():
            tr_f = HotPathTrainer(model_f, schedule=schedule, distributed=with_reducer, det_loss=model_f.det_loss)
        k_f = min(args.steps, 20)
        dt_f, ms_f, _, _ = timed_run(tr_f._step, k_f, 32, tr_f._step.end_stream)
        dt_f = max_over_ranks(dt_f, dist, device)
        tr_f.finish()
        extras["with_all_heads"] = {"scenes_per_s": round(bs * world * k_f / dt_f, 2), "ms_per_step": round(1e3 * dt_f / k_f, 3), "steps": k_f,
                                    "what": "the reference's whole training loss: RPN (BaseBEVBackbone + AnchorHeadSingle) -> proposals (rotated NMS, 9000 -> 512) "
                                            "-> ROI targets (128 sampled rois per scene) -> ConvHead (roi_conv_pool over raw points, occupancy points and "
                                            "x_combine, 6912-cell micro-scene sparse pyramid) -> rcnn cls / reg / corner loss, + occupancy loss"}

    result = None
    if rank == 0:
        scenes = bs * world * args.steps
        heads_txt = {"standin": "(+L2 stand-in for the BEV heads and the ROI head)",
                     "rpn": "-> BaseBEVBackbone -> AnchorHeadSingle + RPN loss (+L2 stand-in for the ROI head)",
                     "full": "-> BaseBEVBackbone -> AnchorHeadSingle + RPN loss -> proposals -> ROI targets -> ConvHead + rcnn loss"}[args.heads]
        result = {
            "metric": "scenes/s fwd+bwd %s bs=2/GPU (BtcDet hot path)" % ("Waymo-shaped synthetic" if waymo else "KITTI-Car"), "value": round(scenes / dt, 3), "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.features == "fp32" else "bf16 operands (activations + weight copies), f32 accumulate / master weights / statistics",
            "data": "synthetic",
            "step_ms": {"p10": round(per_step_ms[int(0.1 * (args.steps - 1))], 3), "median": round(per_step_ms[args.steps // 2], 3),
                        "p90": round(per_step_ms[int(0.9 * (args.steps - 1) + 0.5)], 3), "min": round(per_step_ms[0], 3), "max": round(per_step_ms[-1], 3),
                        "how": "HIP events at the end of every step on the stream its last kernel runs on (rank 0)", "device_allocs_in_timed_region": device_allocs},
            "config": {"workload": ("btcdet_waymo_synth (configs[4] shape) hot path, bs=2/GPU, ~166k pts/scene: HIP voxelize" if waymo else
                                    "btcdet_kitti_car hot path, bs=2/GPU, ~28.6k pts/scene: HIP voxelize") + " (occ+det grids) -> OccTargets3D -> "
                                   "MeanVFE -> VoxelBackBoneDeconv -> OccHead3D+loss -> PassOccVox -> OccVFE -> VoxelBackBone8xOcc -> "
                                   "HeightCompression " + heads_txt + ", fwd+bwd + the reference's optimizer step per parameter group (norm clip 10, "
                                   "decoupled weight decay, Adam, OneCycle lr / beta1), " + ("fp32" if args.features == "fp32" else "bf16 features"),
                       "global_batch": bs * world, "parallelism": "dp%d" % world,
                       "schedule": {"in_order": "in order, one stream (what the reference's own loop gets from these modules)",
                                    "split": "one stream; next batch's weight-independent front on a side stream beside backward; backward in two passes "
                                             "with the detection bucket's all-reduce between them",
                                    "pipelined": "btcdet_amd.trainer.HotPathTrainer 'pipelined': each step prepares ONE batch's weight-independent front "
                                                 "(both voxelizations, occupancy targets, occupancy-branch rulebooks; of the batch after the next one: a loader's "
                                                 "second prefetched batch) on a side stream from a thread of its own; the detection rulebooks are walked by the "
                                                 "occupancy worker right behind PassOccVox; weight gradients on a "
                                                 "side stream, one join per backward; the detection branch's forward on its own stream beside the occupancy "
                                                 "branch's backward; the occupancy bucket's all-reduce (N > 1), the occupancy group's optimizer step and the "
                                                 "NEXT batch's occupancy forward run from a worker thread beside the detection branch's backward, all-reduce "
                                                 "and optimizer step: K steps contain K of everything"}[schedule if step.pipelined or schedule != "pipelined" else "split"],
                       "grad_sync": (None if grad_sync is None else
                                     "btcdet_amd.grad_sync over %s: two flat buckets (occupancy / detection parameters), one all-reduce each per step, "
                                     "launched behind its branch's backward" % grad_sync.transport),
                       "collective": (None if dist is None else {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                                                 "transport": None if grad_sync is None else grad_sync.transport}),
                       "host_cpus": (None if not pinned else "%d CPUs local to the GPU (sysfs local_cpulist), first %d" % (len(pinned), pinned[0])),
                       "streams_on_own_hw_queues": getattr(trainer, "queues_distinct", None), "distinct_batches": nb, "batches": "every priming / warm-up / timed step draws a batch of unseen scenes (seeds bench.rank_seeds)",
                       "points_per_batch": {"min": min(b["n_points"] for b in batches), "max": max(b["n_points"] for b in batches),
                                            "mean": round(sum(b["n_points"] for b in batches) / nb, 1)},
                       "level_rows": level_row_stats(level_rows), "priming_steps": priming, "heads": args.heads},
        }
        result["config"].update(extras)
        straggler = straggler_estimate()
        if straggler is not None:
            result["config"]["predicted_scaling_eff"] = straggler
        if prof is not None:
            summ = prof.summary()
            k = summ.get("conv_apply")
            if k:
                gbs = k["bytes"] / (k["ms"] * 1e-3) / 1e9
                bf = args.features == "bf16"
                tf = k["flops"] / (k["ms"] * 1e-3) / 1e12
                mfma_peak = BF16_MFMA_PEAK_TF if bf else FP32_MFMA_PEAK_TF
                traffic = pmc_traffic_per_launch(waymo, bf)
                avg_s = 1e-3 * k["ms"] / k["launches"]
                # `bound`: algorithmic bytes (every gathered row counts, although most gathers are served by
                # L2 / MALL) against the HBM peak, flops against the dense MFMA peak of the operand type, and -- from the committed PMC
                # pass -- the bytes that really crossed the HBM interface.  If the kernel sits below 0.2 of BOTH real roofs (measured
                # HBM traffic, MFMA flops) neither binds: it is bound by latency (launch + map-tile prologue + per-item barriers,
                # LDS fragment reads), and the JSON says so instead of naming the nearer roof.
                f_hbm, f_mfma = gbs / HBM_PEAK_GBS, tf / mfma_peak
                f_meas = None if traffic is None else traffic / avg_s / 1e9 / HBM_PEAK_GBS
                if (f_meas if f_meas is not None else f_hbm) < 0.2 and f_mfma < 0.2:
                    bound = "latency"
                else:
                    bound = "hbm" if f_hbm >= f_mfma else "mfma"
                result["roofline"] = {"kernel": "conv_apply (fused sparse conv fwd + dgrad, output-stationary MFMA: %s)" % (
                                          "bf16 x bf16 -> f32" if bf else "exact f32 chain, or f32 operands split over the bf16 pipe on the wide layers"),
                                      "bound": bound, "nearer_roof": "hbm" if f_hbm >= f_mfma else "mfma",
                                      "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(f_hbm, 5),
                                      "traffic": traffic, "traffic_source": None if traffic is None else "committed profile (profiles/*_pmc.json: separate "
                                      "rocprofv3 --pmc passes; PMC counters cannot be read inside the timed process), not measured in this run",
                                      "launches_per_step": k["launches"] / prof_steps,
                                      "avg_launch_us": round(1e3 * k["ms"] / k["launches"], 2),
                                      "alg_bytes_per_step": k["bytes"] // prof_steps,
                                      "tflops": round(tf, 3), "mfma_peak_tflops": mfma_peak, "frac_mfma": round(f_mfma, 5),
                                      "frac_hbm_measured": None if f_meas is None else round(f_meas, 5),
                                      "kernel_ms_per_step": round(k["ms"] / prof_steps, 3)}
                # the same algorithmic bytes against the kernels' durations INSIDE the step (committed rocprofv3 statistics of this command)
                sched = scheduled_conv_ms(waymo, bf)
                if sched:
                    bps = k["bytes"] / prof_steps
                    for kind, v in sched.items():
                        result["roofline"]["frac_scheduled" if kind == "serial" else "frac_default_schedule"] = round(bps / (v["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                    result["roofline"]["scheduled_source"] = {kind: v for kind, v in sched.items()}
            for name in ("conv_wgrad", "rulebook"):
                k = summ.get(name)
                if k:
                    result.setdefault("other_kernels", {})[name] = {
                        "GB/s": round(k["bytes"] / (k["ms"] * 1e-3) / 1e9, 2), "ms_per_step": round(k["ms"] / prof_steps, 3),
                        "launches_per_step": k["launches"] / prof_steps, "alg_bytes_per_step": k["bytes"] // prof_steps}
            rb = summ.get("rulebook")
            if rb:
                result["rulebook_hbm_GBps"] = round(rb["bytes"] / (rb["ms"] * 1e-3) / 1e9, 2)
        if world == 1 and not args.no_cpu_baseline and not waymo:
            if pinned and all_cpus:   # the CPU legs (one core; one worker per host core) run unpinned
                os.sched_setaffinity(0, all_cpus)
            result["cpu_baseline"] = cpu_baseline()
        with _StdoutToStderr():   # whatever a library still holds in C stdio's buffer leaves on stderr, not behind the line
            pass
        print(json.dumps(result), flush=True)
    if dist is not None:
        with _StdoutToStderr():
            dist.barrier()
            dist.destroy_process_group()


def level_row_stats(rows):
    """active rows per level over the timed steps (shapes the rulebook walk has read back anyway: measured in the timed run itself)"""
    rows = [r for r in rows if r]
    if not rows:
        return None
    out = {}
    for k in rows[0]:
        v = sorted(r[k] for r in rows if k in r)
        out[k] = {"min": v[0], "median": v[len(v) // 2], "max": v[-1]}
    out["what"] = ("occ_out: rows of the occupancy decoder's output level; det_voxels: detection-grid voxels of the raw points; "
                   "det_voxels_after_pass_occ: + the voxels PassOccVox adds (M''); det_L0..L3 / det_out: the backbone's levels")
    return out


def straggler_estimate():
    """predicted weak-scaling efficiency at 8 ranks from the per-batch step-time distribution measured on ONE GPU
    (tools/straggler.py -> profiles/*_straggler.json): the gradient all-reduce is a barrier, so a step takes as long as the
    slowest of the 8 ranks' batches; E[mean] / E[max of 8 independent draws].  None when the file is absent."""
    try:
        name = next(n for n in ("r06_straggler.json", "r05_straggler.json", "r04_straggler.json", "r03j_straggler.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
        with open(os.path.join(ROOT, "profiles", name)) as f:
            d = json.load(f)
        return {"world": 8, "efficiency": d["predicted_eff_world8"], "source": "committed profile (not measured in this run)",
                "from": "profiles/%s: %d distinct seeded batches, step time "
                "mean %.3f ms, sd %.3f ms" % (name, d["n_batches"], d["mean_ms"], d["sd_ms"])}
    except (OSError, KeyError, ValueError, StopIteration):
        return None


if __name__ == "__main__":
    main()
