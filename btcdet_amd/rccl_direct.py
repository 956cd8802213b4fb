"""A communicator on RCCL itself for the gradient all-reduce of the data-parallel step.

``torch.distributed`` stays the control plane (rendezvous, barriers, broadcasts of the initial parameters: backend "nccl" is
RCCL on ROCm) and carries the unique id of this communicator to the ranks.  The per-step collective does not go through
``ProcessGroupNCCL``: one ``all_reduce`` there costs ~0.17 ms of host time (work object, events, stream bookkeeping -- measured
at world size 1: two of them per step took the data-parallel path from 320-332 to 303-311 scenes/s), and the
step is bound by its host threads.  Here the call is ``ncclAllReduce`` on the caller's HIP stream through ctypes: ordering is
stream ordering, there is no work object, and the host cost is one library call.

The library is the ``librccl.so`` PyTorch itself loads (``torch/lib``), so both communicators share one RCCL instance.

Opt-in (BTC_SYNC_TRANSPORT=rccl), not the default: beside the process group's own RCCL communicator a second one takes further
hardware queues and the step's streams end up sharing theirs -- measured at world size 1, the whole step 422 scenes/s through the
process group, 162 with this communicator merely EXISTING (grad_sync.py docstring).  It pays where the control
plane is not on RCCL (process group on gloo).
"""
import ctypes
import os

import torch

_NCCL_FLOAT32 = 7     # ncclFloat32 (rccl.h ncclDataType_t)
_NCCL_SUM, _NCCL_AVG = 0, 4   # ncclRedOp_t
_UID_BYTES = 128      # NCCL_UNIQUE_ID_BYTES


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * _UID_BYTES)]


_LIB = None


class RcclError(RuntimeError):
    pass


def _lib():
    global _LIB
    if _LIB is None:
        cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so.1", "librccl.so"]
        err = None
        for c in cands:
            try:
                L = ctypes.CDLL(c)
                break
            except OSError as e:
                err = e
        else:
            raise RcclError("librccl.so not found: %r" % (err,))
        L.ncclGetUniqueId.restype = ctypes.c_int
        L.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        L.ncclCommInitRank.restype = ctypes.c_int
        L.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        L.ncclAllReduce.restype = ctypes.c_int
        L.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.ncclCommDestroy.restype = ctypes.c_int
        L.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        L.ncclGetErrorString.restype = ctypes.c_char_p
        L.ncclGetErrorString.argtypes = [ctypes.c_int]
        _LIB = L
    return _LIB


def _check(rc, what):
    if rc != 0:
        raise RcclError("%s failed: %s" % (what, _lib().ncclGetErrorString(rc).decode("utf-8", "replace")))


class RcclComm(object):
    """one RCCL communicator over the ranks of a torch.distributed process group (one process per GPU)"""

    def __init__(self, device, process_group=None):
        import torch.distributed as dist
        L = _lib()
        self.rank = dist.get_rank(process_group)
        self.world = dist.get_world_size(process_group)
        uid = _UniqueId()
        if self.rank == 0:
            _check(L.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        # the id travels over the control plane as a tensor of the group's device type (gloo: host, nccl: device)
        on_dev = dist.get_backend(process_group) == "nccl"
        buf = bytearray(ctypes.string_at(ctypes.addressof(uid), _UID_BYTES)) if self.rank == 0 else bytearray(_UID_BYTES)
        t = torch.frombuffer(buf, dtype=torch.uint8).clone()
        if on_dev:
            t = t.to(device)
        dist.broadcast(t, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0, group=process_group)
        raw = bytes(t.cpu().tolist())
        ctypes.memmove(ctypes.addressof(uid), raw, _UID_BYTES)
        self.device = torch.device(device)
        self._comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(L.ncclCommInitRank(ctypes.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")

    def all_reduce_(self, flat, stream, average=True):
        """in-place sum / mean over the ranks of a contiguous fp32 device tensor, enqueued on `stream` (a torch.cuda.Stream):
        ordered like any kernel on that stream, returns at once"""
        if flat.dtype != torch.float32 or not flat.is_contiguous() or flat.device != self.device:
            raise RcclError("all_reduce_: a contiguous float32 tensor on %s expected" % (self.device,))
        p = flat.data_ptr()
        _check(_lib().ncclAllReduce(p, p, flat.numel(), _NCCL_FLOAT32, _NCCL_AVG if average else _NCCL_SUM, self._comm,
                                    ctypes.c_void_p(stream.cuda_stream)), "ncclAllReduce")

    def destroy(self):
        if self._comm:
            _lib().ncclCommDestroy(self._comm)
            self._comm = ctypes.c_void_p()
