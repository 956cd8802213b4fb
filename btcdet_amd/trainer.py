"""The data-parallel training step of the hot path: the body of the reference's loop
(/root/reference/tools/train_utils/train_utils.py:103-124 -- zero_grad, model_fn, loss.backward, then per optimizer
clip_grad_norm_ / step / lr_scheduler.step) with the schedules this repository measures.

    trainer = HotPathTrainer(model, distributed=dist.is_initialized())      # BtcHotPath (+ optional dense heads)
    for batch, next_batch in pairs(loader):
        loss = trainer.step(batch, next_batch)

Schedules (``schedule=``; config.schedule in bench.py's JSON names what ran):

``in_order``    one stream, one thread, nothing ahead of its turn: prepare -> forward -> backward -> (all-reduce) -> both groups'
                optimizer steps.  What a caller of the modules through the reference's own ``train_one_epoch_multi_opt`` gets
                (bench.py reports it as ``config.in_order_scenes_per_s``).
``split``       one stream; the branches are detached (PASS_GRAD False), so backward runs as two passes -- detection first -- and
                the detection bucket's all-reduce travels during the occupancy branch's backward (round 2's N > 1 schedule).
``pipelined``   (default) the detection branch on its own stream: its forward runs beside the occupancy branch's backward; once
                that backward has returned, a worker thread all-reduces the occupancy bucket, steps the occupancy group, prepares
                the NEXT batch's weight-independent front on the prefetch stream and runs the next batch's occupancy forward --
                with the weights it has just updated -- while this thread runs the detection branch's backward, its bucket's
                all-reduce and the detection group's step.  Every forward pass sees exactly the weights the in-order loop gives it
                (tests/test_hip_prefetch.py); K steps contain K of everything.  With a process group the two collectives of a step
                are issued in a fixed order on every rank (occupancy bucket of step i, detection bucket of step i, occupancy
                bucket of step i + 1, ...: the worker hands over before the training thread launches), on one communicator.
"""
import os
import threading

import torch


_SUMSQ_WS = {}


class MeanSquare2(torch.autograd.Function):
    """ka * mean(a^2) + kb * mean(b^2) over one or two tensors: the L2 stand-in for consumers of the detection branch that are not built
    (the ROI head; without --heads rpn also BEV backbone + anchor head).  GPU tensors: ONE reduction launch forward (fp64 partial sums
    in a fixed order) and ONE elementwise launch backward (csrc/glue.hip btc_sumsq2_*) -- the torch formulation was a norm, two
    multiplications and an add per tensor forward and three multiplications backward, 13 launches of the step's ~400."""

    @staticmethod
    def forward(ctx, a, scale_a, b, scale_b):
        ka = float(scale_a) / max(a.numel(), 1)
        kb = float(scale_b) / max(b.numel(), 1) if b is not None else 0.0
        ctx.k = (ka, kb)
        ok = lambda t: t is None or (t.is_cuda and t.dtype in (torch.float32, torch.bfloat16))
        if not (ok(a) and ok(b)):      # CPU tensors (host-side tests of the schedules): plain torch
            ctx.save_for_backward(a, b)
            ctx.fused = False
            out = torch.linalg.vector_norm(a.reshape(-1), dtype=torch.float32) ** 2 * ka
            return out if b is None else out + torch.linalg.vector_norm(b.reshape(-1), dtype=torch.float32) ** 2 * kb
        from ._lib import check, lib, ptr, stream_ptr
        a = a.contiguous()
        b = b.contiguous() if b is not None else None
        key = (a.device.index, stream_ptr())
        ws = _SUMSQ_WS.get(key)
        if ws is None:
            ws = _SUMSQ_WS[key] = torch.zeros(int(lib().btc_sumsq2_ws_bytes()), dtype=torch.uint8, device=a.device)
        out = torch.empty((1,), dtype=torch.float32, device=a.device)
        check(lib().btc_sumsq2_fwd(ptr(a), a.numel(), int(a.dtype == torch.bfloat16), ka, ptr(b), b.numel() if b is not None else 0,
                                   int(b is not None and b.dtype == torch.bfloat16), kb, ptr(out), ptr(ws), ws.numel(), stream_ptr()), "btc_sumsq2_fwd")
        ctx.save_for_backward(a, b)
        ctx.fused = True
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ka, kb = ctx.k
        if not ctx.fused:
            return (a * (g * (2.0 * ka)).to(a.dtype)), None, (b * (g * (2.0 * kb)).to(b.dtype)) if b is not None else None, None
        from ._lib import check, lib, ptr, stream_ptr
        g = g.reshape(1).to(torch.float32).contiguous()
        da = torch.empty_like(a)
        db = torch.empty_like(b) if b is not None else None
        check(lib().btc_sumsq2_bwd(ptr(a), a.numel(), int(a.dtype == torch.bfloat16), 2.0 * ka, ptr(da), ptr(b), b.numel() if b is not None else 0,
                                   int(b is not None and b.dtype == torch.bfloat16), 2.0 * kb, ptr(db), ptr(g), stream_ptr()), "btc_sumsq2_bwd")
        return da, None, db, None


class MeanSquare(object):
    """scale * mean(x^2) of ONE tensor (callers that add their own terms: --heads rpn, tests)"""

    @staticmethod
    def apply(x, scale):
        return MeanSquare2.apply(x, scale, None, 0.0)


def stand_in_det_loss(ret, batch_dict):
    """L2 stand-ins for both consumers of the detection branch (bench.py's headline configuration)"""
    return MeanSquare2.apply(ret["spatial_features"], 1e-3, ret["x_combine"], 1e-3)


_ONES = {}


def _backward(loss, *more):
    """loss.backward() (of one scalar, or of several at once) with the seed gradient taken from a cache: autograd's own `ones_like` is a
    fill launch per call -- two or three a step"""
    seeds = []
    for t in (loss,) + more:
        key = (t.device, t.dtype, tuple(t.shape))
        one = _ONES.get(key)
        if one is None:
            one = _ONES[key] = torch.ones(t.shape, dtype=t.dtype, device=t.device)
        seeds.append(one)
    torch.autograd.backward((loss,) + more, seeds)


def make_step(model, ddp, proc, opts, grad_sync=None, prefetch_stream=None, threaded=True, det_stream=None, opt_stream=None,
              det_loss=None, pipeline=None):
    """one training step of the hot path -> step(batch, next_batch=None) -> detached loss.

    prefetch_stream: the weight-independent front of the NEXT batch (voxelizations, occupancy targets, the occupancy branch's
    rulebooks: BtcHotPath.prepare) runs on that stream beside this batch's backward pass -- the role DataLoader workers play
    for the reference's CPU voxelizer: from a worker thread while the main thread sits in backward (threaded), or from this
    thread once the backward pass is enqueued.  Every step still does exactly one batch's worth of that work.

    det_stream (not under DistributedDataParallel, which wants one backward per forward): the detection branch is detached
    from the occupancy branch (PASS_GRAD False), so the occupancy branch's BACKWARD does not have to wait for the detection
    branch's FORWARD.  The worker thread calls loss_occ.backward() (autograd runs those nodes on the main stream, where
    their forward ran) while this thread runs the detection branch on det_stream.

    grad_sync: a BucketedGradSync.  With `grad_sync.bucket_of = {"occ": i, "det": j}` (HotPathTrainer sets it) the pipelined
    schedule runs under the reducer too: each thread launches and awaits ITS bucket behind its backward pass.  Without that
    attribute the one-stream schedules are used (split_backward: detection bucket between the two backward passes).

    det_loss(ret, batch_dict) -> scalar: the detection branch's loss (default: the L2 stand-ins)."""
    from . import affinity as _aff
    from .spconv import ops as _ops
    det_loss = det_loss or stand_in_det_loss
    pipeline_wanted = pipeline is None or bool(pipeline)
    _placed = [0]

    def _place_threads():
        """the calling (training) thread, autograd's device thread and the runtime's threads on their CPUs (affinity.place_thread; the
        workers place themselves when they start) -- after each of the first three steps (autograd's thread exists after the first backward,
        a communication library's helpers after the first collective) and once more after the 16th"""
        if _placed[0] < 16:
            _placed[0] += 1
            if _placed[0] <= 3 or _placed[0] == 16:
                _aff.place_thread("train")
                _aff.place_other_threads()
    pending = {}
    pool = None
    if (prefetch_stream is not None and threaded) or det_stream is not None:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=1, initializer=_aff.place_thread, initargs=("occupancy" if pipeline_wanted else "prepare",))
    device = next(model.parameters()).device
    split_backward = grad_sync is not None and ddp is model and len(grad_sync.buckets) > 1 and getattr(grad_sync, "split_backward", False)
    bucket_of = getattr(grad_sync, "bucket_of", None) if grad_sync is not None else None

    # bf16 planes of the split-operand kernel's weights: one launch per parameter group right behind its optimizer step
    _split = {}

    def refresh_planes(group):
        if "lists" not in _split:
            _split["lists"] = split_weight_lists(model) if hasattr(model, "occ_modules") else ((None, ()), (None, ()))
        kind, ws = _split["lists"][group]
        if ws:
            from .spconv import ops as _o
            from ._lib import stream_ptr
            (_o.fast().bf16_weights if kind == "bf16" else _o.fast().split_weights)(ws, stream_ptr())

    def prep(next_batch):
        torch.cuda.set_device(device)
        if timing is None:
            return model.prepare(next_batch, stream=prefetch_stream)
        import time
        t0, c0 = time.perf_counter(), time.thread_time()
        out = model.prepare(next_batch, stream=prefetch_stream)
        timing["p_prepare"] = timing.get("p_prepare", 0.0) + time.perf_counter() - t0
        timing["cpu_prep"] = timing.get("cpu_prep", 0.0) + time.thread_time() - c0
        return out

    def occ_backward(loss):
        torch.cuda.set_device(device)
        _backward(loss)

    # pipelined variant of the det_stream schedule: once the occupancy branch's backward has returned, the worker thread goes on
    # -- the occupancy bucket's all-reduce (process group), the occupancy group's optimizer step (its gradients are complete), the
    # next batch's weight-independent front on the prefetch stream and the next batch's OCCUPANCY FORWARD on the main stream with
    # the updated occupancy weights -- while this thread runs the detection branch's backward, its bucket's all-reduce and its
    # optimizer step on det_stream.  Nothing is skipped and nothing uses stale weights: the occupancy branch of step i + 1 needs
    # the occupancy weights of step i (updated before it runs) and batch i + 1; the detection branch of step i + 1 starts behind
    # step i's detection optimizer on det_stream.  Two backward passes never run at the same time (the weight-gradient side
    # stream's join bookkeeping assumes one).
    if pipeline is None:
        pipeline = True
    pipeline = bool(pipeline and det_stream is not None and ddp is model and (grad_sync is None or bucket_of is not None)
                    and prefetch_stream is not None and threaded and len(opts) == 1 and hasattr(opts[0], "groups") and len(opts[0].groups) == 2)
    ahead_occ = {}

    # work done ahead for a batch is filed with the batch OBJECT and claimed by identity: an id() alone can be reused by another batch
    # once the caller has dropped this one
    def _put(slot, batch, value):
        slot.clear()
        slot["batch"], slot["value"] = batch, value

    def _take(slot, batch):
        hit = slot.get("value") if slot.get("batch") is batch else None
        slot.clear()
        return hit
    # the NEXT batch's weight-independent front on a thread of its own: it shares nothing with the occupancy branch's backward /
    # optimizer step / next forward but the worker thread they all used to queue on -- the step is bound by its host threads (measured,
    # BTC_TRAINER_TIMING=1: training thread 4.1 ms and worker 4.45 ms of host time per 4.9 ms step, the front 1.7 ms of the worker's)
    prep_pool = None
    if pipeline:
        from concurrent.futures import ThreadPoolExecutor as _TPE
        prep_pool = _TPE(max_workers=1, initializer=_aff.place_thread, initargs=("prepare",))

    occ_failed = []   # the worker's exception, if its backward / bucket launch raised (looked at by the training thread, below)

    def occ_tail(loss_occ, next_batch, occ_done, prep_future=None):
        import time
        torch.cuda.set_device(device)
        tw = time.perf_counter() if timing is not None else 0.0
        cw = time.thread_time() if timing is not None else 0.0
        try:
            _backward(loss_occ)
            _ops.join_wgrad()            # (no-op: the end-of-pass callback has joined the side stream into this thread's stream)
            if grad_sync is not None:    # issued BEFORE the hand-over: every rank's order is occupancy bucket, then detection bucket
                grad_sync.launch(bucket_of["occ"])
        except BaseException as e:       # recorded BEFORE the hand-over: the training thread must not issue the detection bucket's
            occ_failed.append(e)         # collective when this rank never issued the occupancy bucket's (fixed per-rank order)
            raise
        finally:
            occ_done.set()
        if timing is not None:
            timing["w_occ_backward"] = timing.get("w_occ_backward", 0.0) + time.perf_counter() - tw
            tw = time.perf_counter()
        if grad_sync is not None:
            grad_sync.wait(bucket_of["occ"])
        opts[0].step(groups=[0])         # occupancy group, on the main stream behind its backward (and its all-reduce)
        refresh_planes(0)
        if next_batch is None:
            return None
        bd_next = prep_future.result() if prep_future is not None else model.prepare(next_batch, stream=prefetch_stream)
        if timing is not None:
            timing["w_opt_prepare"] = timing.get("w_opt_prepare", 0.0) + time.perf_counter() - tw
            tw = time.perf_counter()
        out = occ_forward(bd_next)
        if timing is not None:
            timing["w_occ_forward"] = timing.get("w_occ_forward", 0.0) + time.perf_counter() - tw
            timing["cpu_worker"] = timing.get("cpu_worker", 0.0) + time.thread_time() - cw   # CPU seconds of this thread (a phase much longer than its CPU time was waiting: GIL, GPU, another thread)
        return out

    def occ_forward(bd):
        out = model.forward_occ(bd)
        done = torch.cuda.Event()
        done.record()                    # the loss tensor is complete on this (the main) stream
        return out + (done,)

    timing = {} if os.environ.get("BTC_TRAINER_TIMING") == "1" else None   # host seconds per phase of the pipelined step (tools)

    def _mark(name, t0):
        if timing is not None:
            import time
            t = time.perf_counter()
            timing[name] = timing.get(name, 0.0) + t - t0
            return t
        return 0.0

    prep_ahead = {}

    def step_pipelined(batch, next_batch, after_next=None):
        import time
        t = time.perf_counter() if timing is not None else 0.0
        c0 = time.thread_time() if timing is not None else 0.0
        opts[0].zero_grad(set_to_none=True)
        cur = _take(ahead_occ, batch)
        if cur is None:
            bd = _take(pending, batch)
            cur = occ_forward(bd if bd is not None else model.prepare(batch))
        pending.clear()
        bd, loss_occ, tb, inputs_ready, occ_fwd_done = cur
        occ_done = threading.Event()
        # the weight-independent front: of the NEXT batch if nobody has started it yet, and -- given the batch after that -- of THAT one,
        # a step ahead of its turn (the front of batch k + 2 and the worker's occupancy forward of batch k + 1 were one chain,
        # prepare -> forward, 1.65 + 1.73 ms of host time: the longest in the step once the training thread's share had shrunk)
        prep_future = _take(prep_ahead, next_batch) if next_batch is not None else None
        if prep_future is None and prep_pool is not None and next_batch is not None:
            prep_future = prep_pool.submit(prep, next_batch)
        if prep_pool is not None and after_next is not None and after_next is not next_batch:
            _put(prep_ahead, after_next, prep_pool.submit(prep, after_next))
        fut = pool.submit(occ_tail, loss_occ, next_batch, occ_done, prep_future)
        t = _mark("head", t)
        with torch.cuda.stream(det_stream):
            ret, bd = model.forward_det(bd, inputs_ready)
            loss_det = det_loss(ret, bd)
        t = _mark("det_forward", t)
        occ_done.wait()
        if occ_failed:
            # the occupancy bucket was not launched on this rank: launching the detection bucket now would pair it with the other
            # ranks' occupancy bucket (different size) -- an RCCL hang or silently mis-reduced buffers.  Fail here, cleanly.
            err = occ_failed.pop()
            occ_failed.clear()
            raise RuntimeError("the occupancy branch's backward / bucket launch failed on this rank; the detection bucket's all-reduce "
                               "is NOT issued (collective order)") from err
        t = _mark("wait_occ_backward", t)
        with torch.cuda.stream(det_stream):
            _backward(loss_det)
            _ops.join_wgrad()
            t = _mark("det_backward", t)
            if grad_sync is not None:
                grad_sync.launch(bucket_of["det"])
                grad_sync.wait(bucket_of["det"])
            opts[0].step(groups=[1])     # detection group, on det_stream behind its backward (and its all-reduce)
            refresh_planes(1)
            det_stream.wait_event(occ_fwd_done)
            loss_occ.record_stream(det_stream)
            loss = loss_occ.detach() + loss_det.detach()
            model.mark_step_end(stream=det_stream, upto=bd.get("__gen_id__", -1))
        t = _mark("det_optimizer", t)
        _place_threads()
        nxt = fut.result()
        if nxt is not None:
            _put(ahead_occ, next_batch, nxt)
        t = _mark("wait_worker", t)
        if timing is not None:
            timing["n"] = timing.get("n", 0) + 1
            timing["cpu_main"] = timing.get("cpu_main", 0.0) + time.thread_time() - c0
        return loss

    def step(batch, next_batch=None, after_next=None):
        if pipeline:
            return step_pipelined(batch, next_batch, after_next)
        for o in opts:
            o.zero_grad(set_to_none=True)
        bd = _take(pending, batch)
        if bd is None:
            bd = model.prepare(batch)
        ahead = prefetch_stream is not None and next_batch is not None
        fut = None
        if det_stream is not None and ddp is model:
            main = torch.cuda.current_stream()
            bd, loss_occ, tb, inputs_ready = model.forward_occ(bd)
            fut_occ = pool.submit(occ_backward, loss_occ)
            with torch.cuda.stream(det_stream):
                ret, bd = model.forward_det(bd, inputs_ready)
                loss_det = det_loss(ret, bd)
            fut_occ.result()
            if grad_sync is not None:
                grad_sync.launch_ready()  # the occupancy bucket travels during the detection branch's backward
            fut = pool.submit(prep, next_batch) if (ahead and threaded) else None
            with torch.cuda.stream(det_stream):
                _backward(loss_det)
            main.wait_stream(det_stream)
            loss = loss_occ.detach() + loss_det.detach()
        else:
            ret, tb, bd = ddp(bd)
            loss_det = det_loss(ret, bd)
            fut = pool.submit(prep, next_batch) if (ahead and threaded) else None
            if opt_stream is not None:
                with torch.cuda.stream(opt_stream):
                    _backward(loss_det)
                    opts[0].step(groups=[1])
                _backward(ret["loss_occ"])
                loss = ret["loss_occ"].detach() + loss_det.detach()
            elif split_backward:
                # the branches are detached (PASS_GRAD False): two backward passes give the same gradients as one over the sum.
                # The detection bucket (~90 % of the bytes) is packed and all-reduced BETWEEN them, from this thread -- it travels
                # over xGMI while the occupancy branch's backward runs, with no hook in the autograd thread
                _backward(loss_det)
                grad_sync.launch_ready()
                _backward(ret["loss_occ"])
                loss = ret["loss_occ"].detach() + loss_det.detach()
            else:
                _backward(ret["loss_occ"], loss_det)   # (one pass over both roots: the sum itself is only reported)
                loss = ret["loss_occ"].detach() + loss_det.detach()
        if fut is not None:
            _put(pending, next_batch, fut.result())
        _ops.join_wgrad()   # no-op unless weight gradients are still owed (e.g. a backward pass whose end-of-pass callback never ran)
        _place_threads()
        if grad_sync is not None:
            grad_sync.finish()  # all-reduced mean gradients in the buckets (and in param.grad with assign_grads)
        if opt_stream is not None:
            opts[0].step(groups=[0])
            torch.cuda.current_stream().wait_stream(opt_stream)
        else:
            for o in opts:
                o.step()
            if len(opts) == 1 and hasattr(opts[0], "groups") and len(opts[0].groups) == 2:
                refresh_planes(0)
                refresh_planes(1)
        if ahead and not threaded:
            _put(pending, next_batch, model.prepare(next_batch, stream=prefetch_stream))
        model.mark_step_end()
        return loss
    step.timing = timing
    step.end_stream = det_stream if pipeline else None   # where a step's last kernel runs (per-step timing marks)
    step.pipelined = pipeline
    step.pools = [p_ for p_ in (locals().get("pool"), prep_pool) if p_ is not None]
    return step


def split_weight_lists(model):
    """per parameter group (occupancy, detection): ("split" | "bf16", the sparse-conv weights whose operand copies the apply kernels read)
    -- fp32 features: the three bf16 planes of the split-operand kernel (both channel counts multiples of 32); bf16 features
    (FEATURE_DTYPE: bf16): the bf16 copies of the bf16-operand kernel (one side a multiple of 32, the other of 16).  They are refreshed by
    ONE launch right after the group's optimizer step (binding.cpp split_weights / bf16_weights) instead of one launch per layer at the
    layer's next forward -- ~20 launches per step for a bf16 model.  (None, ()) when the kernels are switched off."""
    from . import _lib
    from .spconv import ops
    from .spconv.conv import SparseConvolution
    F = ops.fast()
    L = _lib.lib()

    def pick(root):
        if F is None or not hasattr(F, "split_weights"):
            return None, ()
        convs = [m for m in root.modules() if isinstance(m, SparseConvolution) and m.weight.is_cuda and m.weight.dtype == torch.float32 and m.weight.requires_grad]
        bb = getattr(root, "backbone_3d", None)
        if getattr(bb, "feature_dtype", None) == torch.bfloat16:
            if L.btc_tune_value(8) == 1 or not hasattr(F, "bf16_weights"):      # BTC_TUNE_BF16_OPERANDS = 1: fp32 weights under bf16 activations
                return None, ()
            ok = lambda m: (m.in_channels % 32 == 0 and m.out_channels % 16 == 0) or (m.out_channels % 32 == 0 and m.in_channels % 16 == 0)
            return "bf16", [m.weight for m in convs if ok(m)]
        if L.btc_tune_value(14) == 1:                                               # BTC_TUNE_SPLIT = 1: the exact kernels everywhere
            return None, ()
        return "split", [m.weight for m in convs if m.in_channels % 32 == 0 and m.out_channels % 32 == 0]
    return pick(model.occ_modules), pick(model.det_modules)


def reference_groups(model, world=1, epochs=40, frames=3712, batch_size=2):
    """the two parameter groups of the reference's two optimizers with the yaml's OPTIMIZATION / OCC_OPTIMIZATION constants
    (btcdet_kitti_car.yaml:331-372: adam_onecycle, norm clip 10, OneCycle over `epochs` x `frames` KITTI training frames)
    -> (groups for GroupOptimizer in the order [occupancy, detection], total optimizer steps)"""
    occ = [p for p in model.occ_modules.parameters() if p.requires_grad]
    det = [p for p in model.det_modules.parameters() if p.requires_grad]
    sched = dict(grad_norm_clip=10.0, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4, lr_clip=1e-7)
    groups = [dict(params=occ, lr=0.003, weight_decay=0.001, **sched), dict(params=det, lr=0.01, weight_decay=0.01, **sched)]
    # len(train_loader) of the reference: DistributedSampler pads every rank to ceil(frames / world) samples, the DataLoader
    # (drop_last False) then yields ceil(that / batch_size) batches (tools/train.py:111-123, build_dataloader)
    per_rank = -(-frames // world)
    return groups, epochs * (-(-per_rank // batch_size))


class HotPathTrainer(object):
    """the step above as an object: owns the optimizer (GroupOptimizer over the reference's two groups), the gradient reducer
    when a process group exists, the streams and the worker thread of the chosen schedule."""

    def __init__(self, model, groups=None, total_steps=None, schedule=None, distributed=None, det_loss=None, process_group=None,
                 optimizer=None, reserve_bytes=None, sync_bn=False):
        import torch.distributed as dist
        from .spconv import ops
        from .train_step import GroupOptimizer
        self.model = model
        self.device = next(model.parameters()).device
        # The step's buffers follow the scene (row counts per level move with every batch, and with what the occupancy head currently
        # predicts), so the caching allocator keeps meeting new maxima and calls hipMalloc -- a device-wide stall -- in the middle of
        # steady-state steps.  One large block, allocated and released here, leaves the allocator a pool it can carve those from.
        # (The allocator's pools are per stream: the block is taken on each stream the schedule allocates on.)
        if reserve_bytes is None:
            reserve_bytes = 512 << 20
        self._reserve_bytes = reserve_bytes if self.device.type == "cuda" else 0
        if distributed is None:
            distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if distributed else 1
        schedule = schedule or os.environ.get("BTC_SCHEDULE", "pipelined")
        if schedule not in ("in_order", "split", "pipelined"):
            raise ValueError("schedule must be in_order, split or pipelined, got %r" % (schedule,))
        self.schedule = schedule
        if optimizer is None:
            if groups is None:
                groups, auto_total = reference_groups(model, self.world)
                total_steps = total_steps or auto_total
            optimizer = GroupOptimizer(groups, total_steps)
        self.optimizer = optimizer
        self.grad_sync = None
        self.sync_bn = 0
        if distributed and sync_bn:
            # --sync_bn of the reference (tools/train.py:32,130-131): every BatchNorm1d over sparse features takes its batch statistics
            # over all ranks' rows (spconv/fused_bn.py); the conv -> BatchNorm fusion and the compiled layer chains step aside for
            # these modules (one all_gather forward and one all_reduce backward per layer, as torch.nn.SyncBatchNorm).  Both
            # branches issue collectives from their own threads under the pipelined schedule, so sync_bn runs in order.
            from .spconv import fused_bn
            self.sync_bn = fused_bn.convert_sync_batchnorm(model, process_group)
            if self.sync_bn and schedule == "pipelined":
                schedule = self.schedule = "split"
        if distributed:
            from .grad_sync import BucketedGradSync
            # `src` of dist.broadcast is a GLOBAL rank: member 0 of a sub-group is not global rank 0
            src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
            for p in model.parameters():   # same start on every rank (DistributedDataParallel does this broadcast in its constructor)
                dist.broadcast(p.data, src=src, group=process_group)
            for b in model.buffers():
                dist.broadcast(b.data, src=src, group=process_group)
            gs = [g["params"] for g in optimizer.groups]
            self.grad_sync = BucketedGradSync([(gs[0], None), (gs[1], None)], process_group=process_group, assign_grads=False)
            self.grad_sync.bucket_of = {"occ": 0, "det": 1}
            self.grad_sync.split_backward = schedule == "split"
            if schedule == "pipelined":
                # each thread launches its bucket at the END of its own backward pass and awaits it at once: a communication stream
                # would overlap nothing and costs two more event hops per bucket.  Pack and all-reduce go on the thread's own stream.
                self.grad_sync.use_comm_stream = False
            if schedule == "split":    # the detection bucket goes first there (launch_ready between the two backward passes)
                self.grad_sync.buckets.reverse()
                self.grad_sync.bucket_of = {"occ": 1, "det": 0}
            optimizer.read_grads_from(self.grad_sync.view_of, self.grad_sync.has_grad, self.grad_sync.missing)
        ops.set_defer_wgrad_join(os.environ.get("BTC_DEFER_WGRAD", "1") != "0")
        cuda = self.device.type == "cuda"
        # the schedule's streams, each on a command-processor pipe of its own (btcdet_amd/streams.py: which of them the runtime had dealt one
        # pipe decided between 3.6 and 5.5 ms per step): detection chain, the next batch's front, the weight gradients' side stream
        self.prefetch_stream = self.det_stream = None
        self.queues_distinct = None
        if cuda:
            from . import _lib
            from .streams import distinct_stream
            used, ok = [torch.cuda.current_stream(self.device)], True
            if schedule == "pipelined":
                self.det_stream, o = distinct_stream(used, self.device)
                used.append(self.det_stream)
                ok = ok and o
            if schedule != "in_order":
                self.prefetch_stream, o = distinct_stream(used, self.device, priority=-1)
                used.append(self.prefetch_stream)
                ok = ok and o
            if _lib.fast() is not None:
                side, o = distinct_stream(used, self.device)
                self._side_stream = side          # (the binding wraps the raw handle: the torch object must outlive it)
                _lib.fast().set_side_stream(side.cuda_stream, self.device.index if self.device.index is not None else torch.cuda.current_device())
                ok = ok and o
            self.queues_distinct = ok
        if self.det_stream is not None:
            # The detection branch's rulebook walk beside its first stage (a fifth active stream) buys nothing once the whole branch runs
            # beside the occupancy backward, and costs a lot: round 4, same box, distinct batches -- 6.1-6.5 ms per step with it against
            # 4.4 ms without.  Part of the schedule, so it is set here, on this model's backbone (not process-wide).
            dbb = getattr(getattr(model, "det_modules", None), "backbone_3d", None)
            if dbb is not None and hasattr(dbb, "walk_async"):
                dbb.walk_async = False
        self._step = make_step(model, model, model.dataset.data_processor, [optimizer], self.grad_sync, self.prefetch_stream,
                               threaded=True, det_stream=self.det_stream, det_loss=det_loss, pipeline=schedule == "pipelined")
        self.end_stream = self._step.end_stream
        if self._reserve_bytes > 0:
            for stream in (torch.cuda.current_stream(self.device), self.det_stream, self.prefetch_stream):
                if stream is not None:
                    with torch.cuda.stream(stream):
                        block = torch.empty(self._reserve_bytes, dtype=torch.uint8, device=self.device)
                        del block

    def step(self, batch, next_batch=None, after_next=None):
        """one optimizer step on `batch`; next_batch (optional) lets the schedule prepare / start it ahead, after_next (optional, the batch
        behind that: a loader's second prefetched batch) lets the pipelined schedule run the weight-independent front two batches
        ahead -- every step still prepares exactly one batch"""
        return self._step(batch, next_batch, after_next)

    def broadcast_buffers(self, src_member=0):
        """BatchNorm running statistics (every module buffer) of group member `src_member` -> all ranks.  The reference trains under
        DistributedDataParallel with the default broadcast_buffers=True (tools/train.py:166-168): rank 0's buffers are re-broadcast at
        every forward, so any rank's checkpoint holds rank 0's running statistics.  Here the buffers stay rank-local during training
        (training-mode BatchNorm does not read them; with sync_bn they are identical anyway) and are aligned when somebody is about
        to read them: call this before checkpoint_state_mult_opt / an evaluation pass.  No-op without a process group."""
        import torch.distributed as dist
        if self.world == 1 or not (dist.is_available() and dist.is_initialized()):
            return 0
        group = self.grad_sync.group if self.grad_sync is not None else None
        src = dist.get_global_rank(group, src_member) if group is not None else src_member
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        via_host = dist.get_backend(group) == "gloo" and self.device.type == "cuda"
        n = 0
        for b in self.model.buffers():
            if via_host:
                h = b.detach().cpu()
                dist.broadcast(h, src=src, group=group)
                b.data.copy_(h)
            else:
                dist.broadcast(b.data, src=src, group=group)
            n += 1
        return n

    def checkpoint_state(self, epoch=None, it=None):
        """the reference's checkpoint dictionary (tools/train_utils/train_utils.py:272-317) of this trainer's model and optimizer, with
        rank 0's BatchNorm buffers on every rank (broadcast_buffers)"""
        from .checkpoint import checkpoint_state_mult_opt
        self.broadcast_buffers()
        return checkpoint_state_mult_opt(self.model, [self.optimizer], epoch=epoch, it=it if it is not None else self.optimizer.iteration)

    def finish(self):
        """end of training: drain the streams and stop the schedule's host threads (the trainer is not usable afterwards)"""
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        for pool in getattr(self._step, "pools", ()):
            pool.shutdown(wait=True)
