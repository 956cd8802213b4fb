"""Gradient all-reduce for the data-parallel step: one process per GPU, RCCL over xGMI.

The reference wraps the model in DistributedDataParallel (tools/train.py:166-168, find_unused_parameters=False with its two
optimizers).  On this path DDP's generality costs ~0.9 ms of a step (one bucket-view copy kernel per parameter, ~120 of them,
plus per-parameter hooks), so the step uses this reducer instead -- same result, the mean of the ranks' gradients:

* parameters are grouped into buckets (here: detection-branch modules, occupancy-branch modules -- the two parameter groups
  of the reference's optimizers); each bucket owns one flat fp32 buffer (+ two trailing elements, see below);
* ``launch(bucket)`` packs the bucket (ONE launch through a pointer table, csrc/optim.hip grads_pack) and enqueues ONE
  all-reduce on a communication stream that waits for the compute stream's position and for the weight gradients still in
  flight on the wgrad side stream -- the compute stream itself goes on (the other branch's backward, the next forward);
* ``wait(bucket)`` makes the current stream wait for that collective; the optimizer then reads the reduced gradients from the
  bucket's slices (``view_of``; ``assign_grads=True`` additionally points every ``param.grad`` at its slice);
* ``finish()`` = launch whatever has not been launched + wait for everything (the one-call form after a single backward).

Transports (``transport=`` / BTC_SYNC_TRANSPORT): ``torch`` -- ``dist.all_reduce`` of the process group (backend "nccl" = RCCL over
xGMI; the default), ``rccl`` -- ncclAllReduce(ncclAvg) on a communicator of our own (btcdet_amd/rccl_direct.py), ``host`` -- reduce
a host copy over gloo (ranks sharing one GPU in the functional tests; the way DDP treats gloo).  Measured at world size 1 on one
MI355X (round 5, pipelined schedule, scenes/s): no process group 445, ``torch`` 422, ``rccl`` 162 -- the call itself is
cheaper there (4 us of host time and an 11 us kernel per 10 MB bucket against 18 us / 46 us), but a SECOND RCCL
communicator in the process takes hardware queues of its own and the step's streams end up sharing theirs (the same run with the
collective itself skipped: 156), so it is opt-in, for a process whose control plane is not on RCCL.

Rank consistency (the contract of DDP with find_unused_parameters=False): every rank must produce a gradient for the same
parameters.  The optimizer treats EVERY parameter of a reduced bucket as present -- a rank that had no gradient for one
contributes zeros to the mean -- so all ranks apply the same update by construction; the two trailing elements of the flat
buffer carry c and c^2, c = the number of locally missing gradients, through the same collective, are read back asynchronously
and checked one step later: a non-zero variance of c over the ranks raises on EVERY rank (DDP raises in the same situation).
(A parameter that is unused on all ranks alike is legal; it then receives an exactly-zero gradient rather than being skipped.)

``param.grad`` (assign_grads=False) keeps the LOCAL gradient and must not be read on the compute stream before ``wait`` /
``finish``: its last writers run on the weight-gradient side stream, which is joined into the communication stream only.
"""
import os

import torch
import torch.distributed as dist

_PACK_KERNEL = True              # one-launch pack (csrc/optim.hip) instead of a multi-tensor copy


class GradSyncError(RuntimeError):
    pass


class _Bucket(object):
    def __init__(self, params, trigger):
        self.params = [p for p in params if p.requires_grad]
        self.trigger = trigger
        self.n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        # [gradients | c, c^2] with c = the number of gradients this rank did not have: one collective carries both, and the mean
        # of c^2 minus the squared mean of c is the variance of c over the ranks -- every rank computes the same value
        self.flat = torch.zeros((self.n + 2,), dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.work = None
        self.launched = False
        self.on_comm = False     # the collective of this step runs on the communication stream
        self.missing = ()        # ids of the parameters without a LOCAL gradient in the step last launched
        self.dirty_tail = False  # the trailing element was written locally (it is zero otherwise and stays zero through a mean)
        self.tables = None       # chunk tables of csrc/optim.hip (one-launch pack), built on first use on a GPU
        self.check = None        # (host tensor, event or None, step): the deferred rank-consistency check
        self.step = 0


class BucketedGradSync(object):
    def __init__(self, buckets, process_group=None, assign_grads=True, transport=None):
        """buckets: list of (parameters, trigger parameter or None).  A trigger is a parameter OUTSIDE the bucket whose
        gradient is accumulated after all of the bucket's (a post-accumulate hook launches the bucket from the autograd
        thread; the training loops here call launch() themselves between their two backward passes instead).
        assign_grads=False: param.grad is left alone -- the optimizer reads the reduced gradients from view_of(param)."""
        self.assign_grads = assign_grads
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        backend = dist.get_backend(process_group)
        self.buckets = [_Bucket(list(params), trig) for params, trig in buckets]
        dev = next((b.flat.device for b in self.buckets if b.params), torch.device("cpu"))
        want = transport or os.environ.get("BTC_SYNC_TRANSPORT") or ("host" if backend == "gloo" and dev.type == "cuda" else "torch")
        self.comm = None
        if want == "rccl":
            try:
                from .rccl_direct import RcclComm
                self.comm = RcclComm(dev, process_group)
            except Exception as e:   # the collective then goes through the process group (still RCCL): slower host side, same result
                import sys
                print("btcdet_amd.grad_sync: direct RCCL communicator unavailable (%r); using torch.distributed" % (e,), file=sys.stderr)
                want = "torch"
        self.transport = want
        self.stage_on_host = want == "host"
        # RCCL averages in the collective (ncclAvg); elsewhere sum, then one scaling launch per bucket
        self.reduce_op = dist.ReduceOp.AVG if backend == "nccl" else dist.ReduceOp.SUM
        self.use_comm_stream = True
        self.check_consistency = True
        self._cs = {}
        self._handles = []
        for b in self.buckets:
            if b.trigger is not None:
                self._handles.append(b.trigger.register_post_accumulate_grad_hook(lambda p, b=b: self.launch(b, only_if_complete=True)))

    # ------------------------------------------------------------------------------------------------------------ launch
    def launch(self, b, only_if_complete=False):
        """pack bucket `b` (a _Bucket or its index) and start its all-reduce; only_if_complete: do nothing unless every
        gradient of the bucket exists already (the early, opportunistic launch)"""
        if isinstance(b, int):
            b = self.buckets[b]
        if b.launched or not b.params:
            return
        grads = [p.grad for p in b.params]
        n_missing = sum(g is None for g in grads)
        if only_if_complete and n_missing:
            return  # not complete yet: finish() will send it
        self._deferred_check(b)
        b.launched = True
        b.step += 1
        if n_missing:   # a conditionally used parameter: legal only if it is missing on every rank (checked one step late)
            b.missing = {id(p) for p, g in zip(b.params, grads) if g is None}
            have = [(v, g) for v, g in zip(b.views, grads) if g is not None]
        else:
            b.missing = ()
            have = list(zip(b.views, grads))
        on_gpu = b.flat.is_cuda
        b.on_comm = on_gpu and not self.stage_on_host and self.use_comm_stream
        if b.on_comm:
            # pack + all-reduce on the communication stream: it waits for what the compute stream has enqueued so far (the
            # gradients dgrad / BatchNorm produced) and for the weight gradients still in flight on the wgrad side stream
            # (ops.join_wgrad joins into the CURRENT stream) -- the compute stream itself is not held up
            cs = self._comm_stream(b.flat.device)
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                from .spconv import ops
                ops.join_wgrad()
                self._pack(b, have, grads, n_missing)
                self._all_reduce(b, cs)
                self._read_tail(b, cs)
            return
        if on_gpu:
            from .spconv import ops
            ops.join_wgrad()  # weight gradients may still be in flight on the side stream (ops.set_defer_wgrad_join)
        self._pack(b, have, grads, n_missing)
        if self.stage_on_host and on_gpu:
            host = b.flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            host.mul_(1.0 / self.world)
            b.flat.copy_(host)
            b.check = (host[-2:].clone(), None, b.step)
        else:
            self._all_reduce(b, torch.cuda.current_stream() if on_gpu else None)
            self._read_tail(b, torch.cuda.current_stream() if on_gpu else None)

    def _all_reduce(self, b, stream):
        if self.comm is not None:
            self.comm.all_reduce_(b.flat, stream, average=True)
        else:
            # a synchronous call runs the collective ON the current stream -- the thread's own, or the communication stream the pack went to
            # (ProcessGroupNCCL, PyTorch >= 2.7); async_op=True would take the process group's own stream: two more event hops per bucket and
            # one more busy stream than the command processor has pipes (btcdet_amd/streams.py).  Host-side it returns as soon as the
            # collective is enqueued either way.  (CPU tensors, gloo: asynchronous, waited for in wait().)
            b.work = dist.all_reduce(b.flat, op=self.reduce_op, group=self.group, async_op=not b.flat.is_cuda)

    def _read_tail(self, b, stream):
        """the reduced count of missing gradients -> pinned host memory, asynchronously; looked at when the bucket is next launched"""
        if not self.check_consistency:
            return
        if b.flat.is_cuda:
            if b.work is not None:   # torch transport: the collective runs on the process group's stream
                b.work.wait()
                b.work = None
            host = self._pinned(b)
            host.copy_(b.flat[-2:], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
            b.check = (host, ev, b.step)
        else:
            if b.work is not None:
                b.work.wait()
                b.work = None
            b.check = (b.flat[-2:].clone(), None, b.step)

    def _pinned(self, b):
        ring = b.__dict__.get("_pin")
        if ring is None:
            ring = b._pin = torch.zeros((4, 2), dtype=torch.float32).pin_memory()
        return ring[b.step % 4]

    def _deferred_check(self, b):
        """raise if, in the step this bucket was last reduced, some rank had no gradient for one of its parameters while this
        one (or another) did: the ranks have then applied different updates (see the module docstring)"""
        chk, b.check = b.check, None
        if chk is None:
            return
        host, ev, step = chk
        if ev is not None:
            ev.synchronize()   # recorded a whole step ago
        m1, m2 = float(host[0]), float(host[1])
        if m1 != 0.0 or m2 != 0.0:
            # some rank had gaps in that step: the REDUCED tail in this rank's flat buffer is non-zero whether or not this rank wrote
            # one locally -- it must not travel into the next all-reduce (matters when the error below is caught by the caller)
            b.dirty_tail = True
        if self.reduce_op != dist.ReduceOp.AVG and not self.stage_on_host and self.comm is None:
            m1, m2 = m1 / self.world, m2 / self.world    # (sum transport: scaled in wait(), after these values were read)
        if m2 - m1 * m1 > 1e-3 * max(1.0, m2):   # variance over the ranks of the number of missing gradients: the same number on every rank
            names = "the mean over ranks of the number of missing gradients was %.3f, of its square %.3f" % (m1, m2)
            raise GradSyncError("data-parallel step %d: ranks disagree on which parameters received a gradient (%s); the reducer needs "
                                "rank-consistent parameter use, like DistributedDataParallel(find_unused_parameters=False) "
                                "(tools/train.py:166-168)" % (step, names))

    def _pack(self, b, have, grads, n_missing):
        """flat bucket <- gradients.  All present, fp32, contiguous, on the GPU and the compiled binding there: ONE launch through a
        pointer table in the kernel arguments (csrc/optim.hip grads_pack); otherwise a multi-tensor copy over two ~120-tensor lists."""
        if n_missing:  # a parameter without gradient this step contributes zeros
            b.flat.zero_()
            b.flat[-2:].copy_(torch.tensor([float(n_missing), float(n_missing) ** 2], dtype=torch.float32), non_blocking=False)
            b.dirty_tail = True
        elif b.dirty_tail or not self.check_consistency:
            # (check_consistency=False: nobody reads the reduced tail back, so nobody learns that another rank made it non-zero -- zero it
            # every step, 8 bytes)
            b.flat[-2:].zero_()
            b.dirty_tail = False
        if not have:
            return
        if b.flat.is_cuda and not n_missing and _PACK_KERNEL:
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
            from .streams import distinct_stream   # (one whose launches overlap with the compute stream's)
            st = self._cs[device.index] = distinct_stream([torch.cuda.current_stream(device)], device)[0]
        return st

    def launch_ready(self):
        """launch every bucket whose gradients are all present (a training loop that runs the branches' backward passes one
        after the other calls this in between)"""
        for b in self.buckets:
            self.launch(b, only_if_complete=True)

    # -------------------------------------------------------------------------------------------------------------- wait
    def wait(self, b):
        """the current stream waits for bucket `b`'s collective (launching it first if nobody has); afterwards the bucket's
        slices hold the mean gradient"""
        if isinstance(b, int):
            b = self.buckets[b]
        if not b.params:
            return
        self.launch(b)
        if b.work is not None:
            b.work.wait()
            b.work = None
        if b.on_comm:   # also in a dry run: the pack and the side-stream join happened over there
            torch.cuda.current_stream().wait_stream(self._comm_stream(b.flat.device))
        if self.comm is None and not self.stage_on_host and self.reduce_op != dist.ReduceOp.AVG:
            b.flat.mul_(1.0 / self.world)
        if self.assign_grads:
            for p, v in zip(b.params, b.views):
                p.grad = v
        b.launched = False

    def finish(self):
        """call after backward, before the optimizer step"""
        for b in self.buckets:
            self.launch(b)
        for b in self.buckets:
            self.wait(b)

    # ------------------------------------------------------------------------------------------------ optimizer interface
    def has_grad(self, param):
        """every parameter of a reduced bucket counts as present on every rank (a rank without a local gradient contributed
        zeros to the mean): the decision is the same everywhere by construction.

        Known divergence from a single-process run (documented): a parameter that is unused on ALL ranks in a step
        -- the one legal case of a missing gradient -- gets a zero-gradient Adam step here (its moments decay, its step counter
        advances, decoupled weight decay applies), while torch.optim.Adam / the reference / this optimizer at N = 1 skip a
        grad-is-None parameter.  Skipping it here too would need the REDUCED count of gaps at optimizer time, i.e. a read-back
        between the all-reduce and the optimizer step of every step; the configured models use every parameter in every step
        (DistributedDataParallel(find_unused_parameters=False) in the reference, tools/train.py:166-168), so the step is not
        paid for.  local_missing() tells a caller that wants the skip which parameters this rank had no gradient for."""
        return True

    def missing(self):
        """ids of the parameters the OPTIMIZER should skip: none (see has_grad); local_missing() has this rank's own gaps"""
        return ()

    def local_missing(self):
        out = set()
        for b in self.buckets:
            out.update(b.missing)
        return out

    def view_of(self, param):
        """the slice of a flat bucket that holds `param`'s reduced gradient after wait() / finish()"""
        m = self.__dict__.get("_view_map")
        if m is None:
            m = self._view_map = {id(p): v for b in self.buckets for p, v in zip(b.params, b.views)}
        return m[id(param)]

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
        if self.comm is not None:
            self.comm.destroy()
            self.comm = None
