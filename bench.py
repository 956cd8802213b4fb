#!/usr/bin/env python
"""bench.py -- BtcDet hot path on MI355X: scenes/s forward+backward, KITTI-Car configuration, bs=2 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the hot path over one synthetic batch whose raw points already sit in HBM:
  GPU voxelization of both grids (cylinder occupancy grid + Cartesian detection grid)
  -> OccTargets3D -> MeanVFE -> VoxelBackBoneDeconv -> OccHead3D (+ occupancy loss) -> PassOccVox
  -> OccVFE -> VoxelBackBone8xOcc -> HeightCompression (+ an L2 stand-in for the out-of-scope BEV heads)
  -> backward (gradient all-reduce over RCCL when N > 1: btcdet_amd/grad_sync.py, or DDP with BTC_BENCH_SYNC=ddp)
  -> the reference's optimizer step per parameter group (norm clip, decoupled weight decay, Adam, OneCycle: btcdet_amd/train_step.py).
Schedule (make_step; config.schedule in the JSON names what ran): every step also prepares the NEXT batch's weight-independent front
(both voxelizations, occupancy targets, occupancy-branch rulebooks) on a side stream beside its backward pass -- one preparation per
step.  Single process: the detection branch (detached from the occupancy branch, PASS_GRAD False) runs on its own stream -- its
forward beside the occupancy branch's backward, and its backward + optimizer step beside the occupancy group's optimizer step
and the NEXT batch's occupancy-branch forward, which the worker thread launches with the occupancy weights it has just updated.
K timed steps contain K of everything; every forward pass sees the weights the one-stream loop would give it
(tests/test_hip_prefetch.py).  BTC_PIPELINE_OCC=0 / BTC_SPLIT_BACKWARD=0 / BTC_PREFETCH=0 step back to the one-stream, in-order
loop; N > 1 keeps one stream for both branches and overlaps the detection bucket's all-reduce instead.
fp32 throughout (--features bf16: BASELINE.json configs[2]).  Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      the dominant kernel (conv_apply = fused sparse conv fwd/dgrad) timed live with HIP events
  cpu_baseline  the CPU oracle timed on the host cores for a bounded sample of the same workload (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("BTC_BENCH_FORCE_DIST") == "1":
    # HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The step keeps four streams busy
    # (main, weight gradients, next-batch preparation, rulebook lookahead); RCCL's streams take queues of their own, and with
    # the default the weight-gradient stream ends up sharing a queue with the main stream: measured at world size 1 over RCCL
    # 8.3 ms per step with 4 queues, 7.3 ms with 8 (no effect without a process group).  Must be set before HIP initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_MFMA_PEAK_TF = 157.3   # same guide: fp32-input MFMA peak
BF16_MFMA_PEAK_TF = 2500.0  # same guide: dense bf16 MFMA peak (AMD's 5 PF figure is 2:1 sparse)


def pmc_traffic_per_launch():
    """HBM bytes per conv_apply launch from the committed PMC passes (profiles/r02h_pmc.json: rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md); PMC counters cannot be read
    inside the timed process, so this is null when the file is absent"""
    try:
        with open(os.path.join(ROOT, "profiles", "r02h_pmc.json")) as f:
            return json.load(f)["conv_apply"]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


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


def build_batches(n_batches, rank, device, batch_size=2, profile="kitti"):
    from btcdet_amd import synth
    batches = []
    for i in range(n_batches):
        seeds = rank_seeds(rank, i, batch_size)
        kw = {"az_step": float(os.environ["BTC_BENCH_AZ_STEP"])} if os.environ.get("BTC_BENCH_AZ_STEP") else {}  # tiny scenes: host-cost probe
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


class MeanSquare(torch.autograd.Function):
    """scale * mean(x^2): the L2 stand-in for the out-of-scope consumers of the detection branch (BEV backbone + dense head,
    point head), one reduction forward and one elementwise launch backward instead of autograd's pow / mean / mul chain"""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.save_for_backward(x)
        ctx.k = float(scale) / max(x.numel(), 1)
        n = torch.linalg.vector_norm(x.reshape(-1), dtype=torch.float32)
        return n * n * ctx.k

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return (x * (g * (2.0 * ctx.k)).to(x.dtype)), None


def make_step(model, ddp, proc, opts, grad_sync=None, prefetch_stream=None, threaded=True, det_stream=None, opt_stream=None):
    """one training step of the hot path.

    prefetch_stream: the weight-independent front of the NEXT batch (voxelizations, occupancy targets, the occupancy branch's
    rulebooks: BtcHotPath.prepare) runs on that stream beside this batch's backward pass -- the role DataLoader workers play
    for the reference's CPU voxelizer: from a worker thread while the main thread sits in backward (threaded), or from this
    thread once the backward pass is enqueued.  Every step still does exactly one batch's worth of that work.

    det_stream (not under DistributedDataParallel, which wants one backward per forward): the detection branch is detached
    from the occupancy branch (PASS_GRAD False), so the occupancy branch's BACKWARD does not have to wait for the detection
    branch's FORWARD.  The worker thread calls loss_occ.backward() (autograd runs those nodes on the main stream, where
    their forward ran) while this thread runs the detection branch on det_stream; both are chains of small launches that do
    not fill the GPU alone.  The detection branch's backward follows on det_stream, and the main stream joins it before the
    optimizer.

    opt_stream (single GPU, one GroupOptimizer whose groups are [occupancy, detection]): the branches are detached, so two
    backward passes give the same gradients as one over the sum.  The detection branch's pass is called with opt_stream
    current: its nodes still run on the main stream (autograd runs a node where its forward ran), but the end-of-pass
    synchronisation -- the engine's wait for the gradient-producing streams and the join of the weight-gradient side stream
    -- lands on opt_stream, and the detection group's optimizer step follows there, beside the occupancy branch's backward on
    the main stream.  The main stream never waits for the detection branch's weight-gradient tail; it joins opt_stream at
    the end of the step."""
    from btcdet_amd.spconv import ops as _ops
    pending = {}
    pool = None
    if (prefetch_stream is not None and threaded) or det_stream is not None:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=1)
    device = next(model.parameters()).device
    split_backward = grad_sync is not None and ddp is model and len(grad_sync.buckets) > 1 and getattr(grad_sync, "split_backward", False)

    def prep(next_batch):
        torch.cuda.set_device(device)
        return model.prepare(next_batch, stream=prefetch_stream)

    def occ_backward(loss):
        torch.cuda.set_device(device)
        loss.backward()

    # pipelined variant of the det_stream schedule (single process): once the occupancy branch's backward has returned, the worker
    # thread goes on -- the occupancy group's optimizer step (its gradients are complete), the next batch's weight-independent front
    # on the prefetch stream and the next batch's OCCUPANCY FORWARD on the main stream with the updated occupancy weights -- while
    # this thread runs the detection branch's backward and optimizer step on det_stream.  Nothing is skipped and nothing uses stale
    # weights: the occupancy branch of step i + 1 needs the occupancy weights of step i (updated before it runs) and batch i + 1;
    # the detection branch of step i + 1 starts behind step i's detection optimizer on det_stream.  Two backward passes never run at
    # the same time (the weight-gradient side stream's join bookkeeping assumes one).
    pipeline = (det_stream is not None and ddp is model and grad_sync is None and prefetch_stream is not None and threaded
                and len(opts) == 1 and hasattr(opts[0], "groups") and len(opts[0].groups) == 2
                and os.environ.get("BTC_PIPELINE_OCC", "1") != "0")
    ahead_occ = {}

    def occ_tail(loss_occ, next_batch, occ_done):
        torch.cuda.set_device(device)
        try:
            loss_occ.backward()
            _ops.join_wgrad()            # (no-op: the end-of-pass callback has joined the side stream into this thread's stream)
        finally:
            occ_done.set()
        opts[0].step(groups=[0])         # occupancy group, on the main stream behind its backward
        if next_batch is None:
            return None
        return occ_forward(model.prepare(next_batch, stream=prefetch_stream))

    def occ_forward(bd):
        out = model.forward_occ(bd)
        done = torch.cuda.Event()
        done.record()                    # the loss tensor is complete on this (the main) stream
        return out + (done,)

    def step_pipelined(batch, next_batch):
        import threading
        opts[0].zero_grad(set_to_none=True)
        cur = ahead_occ.pop(id(batch), None)
        ahead_occ.clear()
        if cur is None:
            bd = pending.pop(id(batch), None)
            cur = occ_forward(bd if bd is not None else model.prepare(batch))
        pending.clear()
        bd, loss_occ, tb, inputs_ready, occ_fwd_done = cur
        occ_done = threading.Event()
        fut = pool.submit(occ_tail, loss_occ, next_batch, occ_done)
        with torch.cuda.stream(det_stream):
            ret, bd = model.forward_det(bd, inputs_ready)
            loss_det = MeanSquare.apply(ret["spatial_features"], 1e-3) + MeanSquare.apply(ret["x_combine"], 1e-3)
        occ_done.wait()
        with torch.cuda.stream(det_stream):
            loss_det.backward()
            _ops.join_wgrad()
            opts[0].step(groups=[1])     # detection group, on det_stream behind its backward
            det_stream.wait_event(occ_fwd_done)
            loss_occ.record_stream(det_stream)
            loss = loss_occ.detach() + loss_det.detach()
            model.mark_step_end(stream=det_stream, upto=bd.get("__gen_id__", -1))
        nxt = fut.result()
        if nxt is not None:
            ahead_occ[id(next_batch)] = nxt
        return loss

    def step(batch, next_batch=None):
        if pipeline:
            return step_pipelined(batch, next_batch)
        for o in opts:
            o.zero_grad(set_to_none=True)
        bd = pending.pop(id(batch), None)
        if bd is None:
            bd = model.prepare(batch)
        pending.clear()
        ahead = prefetch_stream is not None and next_batch is not None
        if det_stream is not None and ddp is model:
            main = torch.cuda.current_stream()
            bd, loss_occ, tb, inputs_ready = model.forward_occ(bd)
            fut_occ = pool.submit(occ_backward, loss_occ)
            with torch.cuda.stream(det_stream):
                ret, bd = model.forward_det(bd, inputs_ready)
                # L2 stand-ins for the out-of-scope consumers of the detection branch
                loss_det = MeanSquare.apply(ret["spatial_features"], 1e-3) + MeanSquare.apply(ret["x_combine"], 1e-3)
            fut_occ.result()
            if grad_sync is not None:
                grad_sync.launch_ready()  # the occupancy bucket travels during the detection branch's backward
            fut = pool.submit(prep, next_batch) if (ahead and threaded) else None
            with torch.cuda.stream(det_stream):
                loss_det.backward()
            main.wait_stream(det_stream)
            loss = loss_occ.detach() + loss_det.detach()
        else:
            ret, tb, _ = ddp(bd)
            # occupancy loss (real) + L2 stand-ins for the out-of-scope consumers of the detection branch
            loss_det = MeanSquare.apply(ret["spatial_features"], 1e-3) + MeanSquare.apply(ret["x_combine"], 1e-3)
            fut = pool.submit(prep, next_batch) if (ahead and threaded) else None
            if opt_stream is not None:
                with torch.cuda.stream(opt_stream):
                    loss_det.backward()
                    opts[0].step(groups=[1])
                ret["loss_occ"].backward()
                loss = ret["loss_occ"].detach() + loss_det.detach()
            elif split_backward:
                # the branches are detached (PASS_GRAD False): two backward passes give the same gradients as one over the sum.
                # The detection bucket (~90 % of the bytes) is packed and all-reduced BETWEEN them, from this thread -- it travels
                # over xGMI while the occupancy branch's backward runs, with no hook in the autograd thread
                loss_det.backward()
                grad_sync.launch_ready()
                ret["loss_occ"].backward()
                loss = ret["loss_occ"].detach() + loss_det.detach()
            else:
                loss = ret["loss_occ"] + loss_det
                loss.backward()
        if fut is not None:
            pending[id(next_batch)] = fut.result()
        _ops.join_wgrad()   # no-op unless weight gradients are still owed (e.g. a backward pass whose end-of-pass callback never ran)
        if grad_sync is not None:
            grad_sync.finish()  # all-reduced mean gradients in param.grad
        if opt_stream is not None:
            opts[0].step(groups=[0])
            torch.cuda.current_stream().wait_stream(opt_stream)
        else:
            for o in opts:
                o.step()
        if ahead and not threaded:
            pending[id(next_batch)] = model.prepare(next_batch, stream=prefetch_stream)
        model.mark_step_end()
        return loss
    step.end_stream = det_stream if pipeline else None   # where a step's last kernel runs (per-step timing marks)
    return step


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
    ap.add_argument("--workload", choices=["kitti", "waymo"], default="kitti",
                    help="kitti: the configuration BASELINE.json's metric is quoted on (default); waymo: the Waymo-shaped synthetic "
                         "scenes of configs[4] (~166 k points/scene, 1504 x 1504 x 40 grid) -- same path, 6x the work per scene")
    ap.add_argument("--features", choices=["fp32", "bf16"], default="fp32",
                    help="fp32: the reference's precision (default, the headline number); bf16: BASELINE.json configs[2] -- bfloat16 "
                         "activations between sparse layers, fp32 weights / accumulation / statistics")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback for the hot path)"
    # BTC_BENCH_BACKEND=gloo lets N ranks share one GPU (functional check of the N > 1 path on a 1-GPU box); the default
    # is one rank per GPU over RCCL ("nccl" on ROCm)
    backend = os.environ.get("BTC_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # host threads of this rank next to its GPU (btcdet_amd/affinity.py): the step is launch-rate sensitive, and on a two-socket
    # host an unpinned process that lands on the far socket loses 3-6 % (and makes the number box-dependent)
    from btcdet_amd.affinity import pin_to_gpu
    all_cpus = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    pinned = pin_to_gpu(dev_index, local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    dist = None
    # BTC_BENCH_FORCE_DIST=1: take the distributed path (process group, DDP wrapper, barriers) at world size 1 too --
    # the way to exercise the RCCL code path on a single-GPU box
    use_dist = world > 1 or os.environ.get("BTC_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", **({} if os.environ.get("BTC_BENCH_LAZY_NCCL") == "1" else {"device_id": device}))
        else:
            dist.init_process_group(backend=backend)

    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    from btcdet_amd.spconv import ops

    torch.manual_seed(666)
    np.random.seed(666 + rank)
    waymo = args.workload == "waymo"
    cfg = load_cfg(os.path.join(ROOT, "btcdet_amd", "cfgs", "btcdet_waymo_synth.yaml") if waymo else None)
    if args.features == "bf16":
        cfg.MODEL.OCC.BACKBONE_3D["FEATURE_DTYPE"] = "bf16"
        cfg.MODEL.BACKBONE_3D["FEATURE_DTYPE"] = "bf16"
    model = BtcHotPath(cfg, device=device).to(device)
    model.train()
    ddp, grad_sync = model, None
    occ_params = [p for p in model.occ_modules.parameters() if p.requires_grad]
    det_params = [p for p in model.det_modules.parameters() if p.requires_grad]
    if use_dist and os.environ.get("BTC_BENCH_SYNC", "bucketed") == "ddp":
        # gradient_as_bucket_view: gradients are written straight into the all-reduce buckets; broadcast_buffers=False:
        # BatchNorm running statistics stay rank-local between checkpoints instead of being re-broadcast from rank 0 before
        # every forward (training-mode BN never reads them; the reference's default DDP re-broadcasts, tools/train.py:166-168);
        # static_graph: the same parameters are used every step.
        kw = {"gradient_as_bucket_view": True, "broadcast_buffers": False, "static_graph": True}
        for item in os.environ.get("BTC_DDP_OPTS", "").split(","):  # e.g. gradient_as_bucket_view=1,broadcast_buffers=0
            if "=" in item:
                k, v = item.split("=")
                kw[k] = (float(v) if k == "bucket_cap_mb" else bool(int(v)))
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev_index], find_unused_parameters=False, **kw)
    elif use_dist and os.environ.get("BTC_BENCH_NOSYNC") != "1":
        # default: btcdet_amd/grad_sync.py -- two flat buckets (detection / occupancy parameters), the detection bucket's
        # all-reduce overlapped with the occupancy branch's backward.  Measured at world size 1 over RCCL: DDP (tuned as above)
        # 10.1 ms per step, this reducer see DESIGN.md, no reducer 9.2 ms.
        from btcdet_amd.grad_sync import BucketedGradSync
        for p in model.parameters():  # same start on every rank (DDP does this broadcast in its constructor)
            dist.broadcast(p.data, src=0)
        for b in model.buffers():
            dist.broadcast(b.data, src=0)
        head_param = next(p for p in model.occ_modules.occ_dense_head.parameters() if p.requires_grad)
        mode = os.environ.get("BTC_SYNC_BUCKETS", "split")
        view = os.environ.get("BTC_BENCH_OPTIM", "lean") != "torch"     # the optimizer reads the buckets' slices (no param.grad stores)
        if mode == "early":   # detection bucket launched from a hook in mid-backward (comm hidden, hook's Python in the engine thread)
            grad_sync = BucketedGradSync([(det_params, head_param), (occ_params, None)])
        elif mode == "two":
            grad_sync = BucketedGradSync([(det_params, None), (occ_params, None)])
        elif mode == "one":   # one flat bucket sent after backward: the least host work; ~10 MB of all-reduce exposed
            grad_sync = BucketedGradSync([(det_params + occ_params, None)], assign_grads=not view)
        else:                 # default: two buckets, the detection bucket's all-reduce overlapped with the occupancy branch's backward
            grad_sync = BucketedGradSync([(det_params, None), (occ_params, None)], assign_grads=not view)   # (make_step: split_backward)
            grad_sync.split_backward = True
    # the reference's optimizer step per parameter group (tools/train_utils/train_utils.py:121-124; yaml:331-372): gradient-norm
    # clip at 10, adam_onecycle = decoupled weight decay + Adam(betas=(mom, 0.99)) with lr / mom on the OneCycle schedule of a
    # 40-epoch run over KITTI's 3712 training frames -- btcdet_amd/train_step.py (checked against the reference's own
    # OptimWrapper / OneCycle, tests/test_train_step_cpu.py).  The two optimizers are the two groups of one object: same
    # arithmetic per group, one Python call.
    # weight gradients on a side stream for the whole backward pass, joined once at its end (not under DDP, whose hooks read
    # them in mid-backward)
    ops.set_defer_wgrad_join(ddp is model and os.environ.get("BTC_DEFER_WGRAD", "1") != "0")
    from btcdet_amd.train_step import GroupOptimizer
    total_steps = 40 * (3712 // (2 * world))
    sched_kw = dict(grad_norm_clip=10.0, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4, lr_clip=1e-7)
    groups = [dict(params=occ_params, lr=0.003, weight_decay=0.001, **sched_kw), dict(params=det_params, lr=0.01, weight_decay=0.01, **sched_kw)]
    if os.environ.get("BTC_BENCH_OPTIM", "lean") == "torch":   # plain torch Adam (no clip / schedule): A-B runs only
        opts = [torch.optim.Adam([{"params": occ_params, "lr": 3e-3}, {"params": det_params, "lr": 1e-2}], betas=(0.9, 0.99), fused=True)]
    else:
        opts = [GroupOptimizer(groups, total_steps)]
        if grad_sync is not None and not grad_sync.assign_grads:
            opts[0].read_grads_from(grad_sync.view_of, grad_sync.has_grad, grad_sync.missing)
    bs = 2
    batches = build_batches(4, rank, device, bs, args.workload)
    # the next batch's weight-independent front runs on a high-priority side stream beside this batch's backward
    prefetch = torch.cuda.Stream(device=device, priority=-1) if os.environ.get("BTC_PREFETCH", "2") != "0" else None
    # the detection branch on its own stream, beside the occupancy branch's backward (make_step)
    # Single process (default): the detection branch's forward runs on its own stream beside the occupancy branch's backward
    # (make_step, det_stream; gradients equal the one-stream schedule's: tests/test_hip_prefetch.py).  Measured back to back:
    # 353 -> 374 scenes/s fp32, 354 -> 412 bf16.  With a gradient reducer the detection bucket (90 % of the bytes) would only be
    # complete at the very end of the step, its all-reduce exposed: 305-314 -> 280-284 at world size 1 over RCCL, so the
    # distributed path keeps the split-backward schedule (BTC_SPLIT_BACKWARD=0/1 overrides either default).
    split_default = "1" if (grad_sync is None and prefetch is not None) else "0"   # (BTC_PREFETCH=0 is the in-order, one-stream schedule)
    det_stream = torch.cuda.Stream(device=device) if (ddp is model and os.environ.get("BTC_SPLIT_BACKWARD", split_default) == "1") else None
    if det_stream is not None and "BTC_DET_WALK_ASYNC" not in os.environ:
        # the detection branch's rulebook walk beside its first stage buys nothing once the whole branch runs beside the occupancy
        # backward (377 vs 374 scenes/s without it): one stream less
        import btcdet_amd.backbones_3d as _bb3d
        _bb3d.DET_WALK_ASYNC = False
    # BTC_EARLY_OPT=1: the detection group's optimizer step beside the occupancy branch's backward (make_step; single process only
    # -- with a gradient reducer the detection bucket's all-reduce takes that slot).  Worth 1.7 % while a group's step was ~10
    # multi-tensor torch ops (333.8 -> 339.5 scenes/s); with the three-launch step of csrc/optim.hip there is nothing left to hide
    # (349.9 without, 348.8 with), so it is off by default.
    early_opt = (grad_sync is None and ddp is model and det_stream is None and isinstance(opts[0], GroupOptimizer)
                 and os.environ.get("BTC_EARLY_OPT", "0") == "1")
    opt_stream = torch.cuda.Stream(device=device) if early_opt else None
    step = make_step(model, ddp, model.dataset.data_processor, opts, grad_sync, prefetch, threaded=os.environ.get("BTC_PREFETCH", "2") == "2",
                     det_stream=det_stream, opt_stream=opt_stream)
    nb = len(batches)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # Setup, before the W warmup steps: the caching allocator, the per-batch-size geometry plans and the flat optimizer buffers reach
    # their steady state only after every one of the 4 synthetic batches has been seen a few times; a short W (the driver's choice)
    # would otherwise leave hipMalloc calls and plan construction inside the timed region (config.priming_steps in the JSON).
    priming = max(0, 16 - args.warmup)
    for i in range(priming):
        step(batches[i % nb], batches[(i + 1) % nb])
    for i in range(priming, priming + args.warmup):
        step(batches[i % nb], batches[(i + 1) % nb])
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    allocs0 = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    marks[0].record(step.end_stream)
    first = priming + args.warmup
    for i in range(first, first + args.steps):  # each step prepares its successor: K steps, K preparations
        step(batches[i % nb], batches[(i + 1) % nb])
        marks[i - first + 1].record(step.end_stream)  # end of the step's work on the stream its last kernel runs on (no host wait)
    sync()
    dt = time.perf_counter() - t0
    per_step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    device_allocs = torch.cuda.memory_stats(device).get("num_device_alloc", 0) - allocs0   # hipMalloc calls inside the timed region
    # roofline leg: the SAME steps once more with a HIP event pair around every sparse-conv / rulebook launch
    # (kept out of the timed region above because counting the pairs of each rulebook needs a read-back)
    prof = None
    prof_steps = 0
    if not args.no_roofline and rank == 0:
        prof = ops.LaunchProfile()
        ops.PROFILE = prof
        prof_steps = min(args.steps, 8)
    if not args.no_roofline:
        # one stream, one thread: the event pairs time kernels that run alone
        plain_step = make_step(model, ddp, model.dataset.data_processor, opts, grad_sync)
        for i in range(min(args.steps, 8)):
            plain_step(batches[i % len(batches)])
        sync()
    ops.PROFILE = None
    dt = max_over_ranks(dt, dist, device)
    if grad_sync is not None and os.environ.get("BTC_SYNC_TIMING") == "1" and rank == 0:
        from btcdet_amd import grad_sync as _gs
        n = max(_gs._TIMING.get("n", 1), 1)
        print("grad_sync host ms per step:", {k: round(v / n * 1e3, 3) for k, v in _gs._TIMING.items() if k != "n"}, file=sys.stderr)

    result = None
    if rank == 0:
        scenes = bs * world * args.steps
        result = {
            "metric": "scenes/s fwd+bwd %s bs=2/GPU (BtcDet hot path)" % ("Waymo-shaped synthetic" if waymo else "KITTI-Car"), "value": round(scenes / dt, 3), "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.features == "fp32" else "bf16 operands (activations + weight copies), f32 accumulate / master weights / statistics",
            "data": "synthetic",
            "step_ms": {"p10": round(per_step_ms[int(0.1 * (args.steps - 1))], 3), "median": round(per_step_ms[args.steps // 2], 3),
                        "p90": round(per_step_ms[int(0.9 * (args.steps - 1) + 0.5)], 3), "min": round(per_step_ms[0], 3), "max": round(per_step_ms[-1], 3),
                        "how": "HIP events at the end of every step on the main stream (rank 0)", "device_allocs_in_timed_region": device_allocs},
            "config": {"workload": ("btcdet_waymo_synth (configs[4] shape) hot path, bs=2/GPU, ~166k pts/scene: HIP voxelize" if waymo else
                                    "btcdet_kitti_car hot path, bs=2/GPU, ~28.6k pts/scene: HIP voxelize") + " (occ+det grids) -> OccTargets3D -> "
                                   "MeanVFE -> VoxelBackBoneDeconv -> OccHead3D+loss -> PassOccVox -> OccVFE -> VoxelBackBone8xOcc -> "
                                   "HeightCompression (+L2 stand-in for the out-of-scope BEV heads), fwd+bwd + the reference's optimizer step per parameter group (norm clip 10, "
                                   "decoupled weight decay, Adam, OneCycle lr / beta1), " + ("fp32" if args.features == "fp32" else "bf16 features"),
                       "global_batch": bs * world, "parallelism": "dp%d" % world,
                       "schedule": (("each step prepares the NEXT batch's weight-independent front (both voxelizations, occupancy targets, "
                                     "occupancy-branch rulebooks) on a side stream beside its backward pass, one preparation per step; "
                                     "weight gradients on a side stream, one join per backward" if prefetch is not None else "in order, one stream")
                                    + ("; the detection branch's forward on its own stream beside the occupancy branch's backward" if det_stream is not None else "")
                                    + ("; the occupancy group's optimizer step and the NEXT batch's occupancy-branch forward (with those updated weights) "
                                       "run from the worker thread beside the detection branch's backward and optimizer step: K steps contain K "
                                       "occupancy forwards, K detection forwards, K backward passes and K optimizer steps per group"
                                       if getattr(step, "end_stream", None) is not None else "")),
                       "grad_sync": ("DistributedDataParallel" if ddp is not model else
                                     (None if grad_sync is None else ("btcdet_amd.grad_sync: detection bucket all-reduced during the occupancy branch's backward, occupancy bucket after it"
                                                                     if getattr(grad_sync, "split_backward", False) else "btcdet_amd.grad_sync: flat bucket(s), all-reduce after backward"))),
                       "collective": (None if dist is None else {"backend": dist.get_backend(), "world_size": dist.get_world_size()}),
                       "host_cpus": (None if not pinned else "%d CPUs local to the GPU (sysfs local_cpulist), first %d" % (len(pinned), pinned[0])),
                       "points_per_batch": [b["n_points"] for b in batches], "priming_steps": priming},
        }
        if prof is not None:
            summ = prof.summary()
            k = summ.get("conv_apply")
            if k:
                gbs = k["bytes"] / (k["ms"] * 1e-3) / 1e9
                bf = args.features == "bf16"
                tf = k["flops"] / (k["ms"] * 1e-3) / 1e12
                mfma_peak = BF16_MFMA_PEAK_TF if bf else FP32_MFMA_PEAK_TF
                traffic = None if (waymo or bf) else pmc_traffic_per_launch()
                avg_s = 1e-3 * k["ms"] / k["launches"]
                # `bound`: the roof the kernel sits closer to.  Algorithmic bytes (every gathered row counts, although
                # most gathers are served by L2 / MALL) against the HBM peak, flops against the dense MFMA peak of the operand type;
                # frac_hbm_measured is what the PMC counters saw actually crossing the HBM interface.
                f_hbm, f_mfma = gbs / HBM_PEAK_GBS, tf / mfma_peak
                result["roofline"] = {"kernel": "conv_apply (fused sparse conv fwd + dgrad, output-stationary MFMA %s)" % ("bf16 x bf16 -> f32" if bf else "f32"),
                                      "bound": "hbm" if f_hbm >= f_mfma else "mfma",
                                      "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(f_hbm, 5),
                                      "traffic": traffic, "launches_per_step": k["launches"] / prof_steps,
                                      "avg_launch_us": round(1e3 * k["ms"] / k["launches"], 2),
                                      "alg_bytes_per_step": k["bytes"] // prof_steps,
                                      "tflops": round(tf, 3), "mfma_peak_tflops": mfma_peak, "frac_mfma": round(f_mfma, 5),
                                      "frac_hbm_measured": None if traffic is None else round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 5),
                                      "kernel_ms_per_step": round(k["ms"] / prof_steps, 3)}
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
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
