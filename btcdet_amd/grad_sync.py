"""Gradient all-reduce for the data-parallel step: one process per GPU, RCCL (``backend="nccl"``) over xGMI.

The reference wraps the model in DistributedDataParallel (tools/train.py:166-168).  On this path DDP's generality costs
~0.9 ms of a 9.2 ms step (one bucket-view copy kernel per parameter, ~120 of them, plus per-parameter hooks), so the
bench uses this reducer instead -- same result, the mean of the ranks' gradients in every ``param.grad``:

* parameters are grouped into buckets (here: detection-branch modules, occupancy-branch modules -- the two parameter groups
  of the reference's optimizers); each bucket owns one flat fp32 buffer;
* a bucket may name a TRIGGER parameter outside itself: when that parameter's gradient is accumulated the bucket is
  launched if all of its own gradients exist (otherwise it waits for ``finish()``).  The detection bucket is triggered by a
  parameter of the occupancy head: the detection branch is detached from the occupancy branch (PASS_GRAD False) and its
  autograd nodes were created later, so the engine runs all of them before the first occupancy node -- the detection
  gradients (~90 % of the bytes) travel while the whole occupancy branch is still in backward.  Launch = one multi-tensor
  copy into the flat buffer + one asynchronous all-reduce;
* ``finish()`` (before the optimizer step) launches whatever has not been launched, waits, scales by 1/world and points
  every ``param.grad`` at its slice of the flat buffer (no copy back).

``BTC_BENCH_SYNC=ddp`` makes bench.py use DistributedDataParallel instead."""
import torch
import torch.distributed as dist


import os
_TIMING = {} if os.environ.get("BTC_SYNC_TIMING") == "1" else None  # host seconds spent in _launch / finish (tools)
_DRYRUN = os.environ.get("BTC_SYNC_DRYRUN") == "1"                   # A-B runs: everything but the collective itself
_PACK_KERNEL = os.environ.get("BTC_SYNC_PACK", "1") != "0"            # one-launch pack (csrc/optim.hip) instead of a multi-tensor copy


class _Bucket(object):
    def __init__(self, params, trigger):
        self.params = [p for p in params if p.requires_grad]
        self.trigger = trigger
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros((n,), dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.work = None
        self.launched = False
        self.missing = ()
        self.tables = None   # chunk tables of csrc/optim.hip (one-launch pack), built on first use on a GPU


class BucketedGradSync(object):
    def __init__(self, buckets, process_group=None, assign_grads=True):
        """buckets: list of (parameters, trigger parameter or None); the trigger is the parameter whose gradient arrives last.
        assign_grads=False: finish() leaves param.grad alone -- the optimizer reads the reduced gradients from view_of(param)
        (saves one attribute store per parameter and step).  In that mode param.grad keeps the LOCAL, un-reduced gradient:
        anything that must see the reduced one (a norm clip, logging) has to read view_of(param) as well."""
        self.assign_grads = assign_grads
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.stage_on_host = dist.get_backend(process_group) == "gloo"  # gloo's device path is very slow: reduce a host copy (as DDP does)
        # RCCL averages in the collective (ncclAvg); elsewhere sum, then one scaling launch per bucket in finish()
        self.reduce_op = dist.ReduceOp.AVG if dist.get_backend(process_group) == "nccl" else dist.ReduceOp.SUM
        self.buckets = [_Bucket(list(params), trig) for params, trig in buckets]
        self._handles = []
        self.use_comm_stream = os.environ.get("BTC_SYNC_COMM_STREAM", "1") != "0"
        self._cs = {}
        for b in self.buckets:
            if b.trigger is not None:
                self._handles.append(b.trigger.register_post_accumulate_grad_hook(lambda p, b=b: self._launch(b)))

    def _launch(self, b, early=True):
        if b.launched or not b.params:
            return
        if _TIMING is not None:
            import time
            t0 = time.perf_counter()
            try:
                return self._launch_impl(b, early)
            finally:
                _TIMING["launch"] = _TIMING.get("launch", 0.0) + time.perf_counter() - t0

        return self._launch_impl(b, early)

    def _launch_impl(self, b, early=True):
        grads = [p.grad for p in b.params]
        if early and any(g is None for g in grads):
            return  # not complete yet: finish() will send it
        b.launched = True
        if None in grads:   # rare: a conditionally used parameter
            b.missing = {id(p) for p, g in zip(b.params, grads) if g is None}
            have = [(v, g) for v, g in zip(b.views, grads) if g is not None]
        else:
            b.missing = ()
            have = list(zip(b.views, grads))
        if b.flat.is_cuda and not self.stage_on_host and self.use_comm_stream:
            # pack + all-reduce on a communication stream: it waits for what the compute stream has enqueued so far (the
            # gradients dgrad / BatchNorm produced) and for the weight gradients still in flight on the wgrad side stream
            # (ops.join_wgrad joins into the CURRENT stream) -- the compute stream itself is not held up, so the next
            # branch's backward keeps running beside the wgrads and the collective
            cs = self._comm_stream(b.flat.device)
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                from .spconv import ops
                ops.join_wgrad()
                self._pack(b, have, grads)
                if not _DRYRUN:
                    b.work = dist.all_reduce(b.flat, op=self.reduce_op, group=self.group, async_op=True)
            return
        if b.flat.is_cuda:
            from .spconv import ops
            ops.join_wgrad()  # weight gradients may still be in flight on the side stream (ops.set_defer_wgrad_join)
        self._pack(b, have, grads)
        if self.stage_on_host and b.flat.is_cuda:
            host = b.flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            b.flat.copy_(host)
        else:
            b.work = dist.all_reduce(b.flat, op=self.reduce_op, group=self.group, async_op=True)

    def _pack(self, b, have, grads):
        """flat bucket <- gradients.  All present, fp32, contiguous, on the GPU and the compiled binding there: ONE launch through a
        pointer table in the kernel arguments (csrc/optim.hip grads_pack); otherwise a multi-tensor copy over two ~120-tensor lists
        (measured at world size 1 over RCCL: 305.7 vs 299.8 scenes/s)."""
        if len(have) != len(grads):  # a parameter without gradient this step contributes zeros
            b.flat.zero_()
        if not have:
            return
        if b.flat.is_cuda and len(have) == len(grads) and _PACK_KERNEL:
            from . import _lib
            F = _lib.fast()
            if F is not None and all(g.dtype == torch.float32 and g.is_contiguous() for g in grads):
                if b.tables is None:
                    from .train_step import chunk_tables
                    b.sizes = [p.numel() for p in b.params]
                    b.tables = chunk_tables(b.sizes, b.flat.device)
                t = b.tables
                F.pack_grads(grads, t["seg"], t["off"], t["len"], t["flat"], t["seg0"], b.sizes, b.flat, _lib.stream_ptr())
                return
        torch._foreach_copy_([v for v, _ in have], [g for _, g in have])

    def _comm_stream(self, device):
        st = self._cs.get(device.index)
        if st is None:
            st = self._cs[device.index] = torch.cuda.Stream(device=device)
        return st

    def launch_ready(self):
        """launch every bucket whose gradients are all present (a training loop that runs the branches' backward passes one
        after the other calls this in between)"""
        for b in self.buckets:
            self._launch(b, early=True)

    def finish(self):
        """call after backward, before the optimizer step"""
        if _TIMING is not None:
            import time
            t0 = time.perf_counter()
            try:
                return self._finish_impl()
            finally:
                _TIMING["finish"] = _TIMING.get("finish", 0.0) + time.perf_counter() - t0
                _TIMING["n"] = _TIMING.get("n", 0) + 1
        return self._finish_impl()

    def _finish_impl(self):
        for b in self.buckets:
            self._launch(b, early=False)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None
            if b.params:
                if self.reduce_op != dist.ReduceOp.AVG or self.stage_on_host:
                    b.flat.mul_(1.0 / self.world)
                if self.assign_grads:
                    for p, v in zip(b.params, b.views):
                        p.grad = v
            b.launched = False

    def has_grad(self, param):
        """whether `param` received a gradient in the step whose buckets were last launched (a parameter that did not
        contributes zeros to the all-reduce; an optimizer should skip it, as torch.optim.Adam skips grad-is-None parameters)"""
        return not any(id(param) in b.missing for b in self.buckets)

    def missing(self):
        """ids of the parameters that had no gradient in the step whose buckets were last launched (normally empty)"""
        out = set()
        for b in self.buckets:
            out.update(b.missing)
        return out

    def view_of(self, param):
        """the slice of a flat bucket that holds `param`'s reduced gradient after finish()"""
        m = self.__dict__.get("_view_map")
        if m is None:
            m = self._view_map = {id(p): v for b in self.buckets for p, v in zip(b.params, b.views)}
        return m[id(param)]

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
