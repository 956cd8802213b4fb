"""Operator layer: rulebook construction and the autograd functions over libbtcdet_hip.so.

Mirrors ``spconv.ops`` / ``spconv.functional`` of spconv v1.2.1: the
reference reaches these through SubMConv3d / SparseConv3d / SparseConvTranspose3d /
SparseInverseConv3d / SparseMaxPool3d (/root/reference/btcdet/models/backbones_3d/
spconv_backbone.py:12-29).  The rulebook is kept as two dense neighbour maps (see
include/btcdet_hip.h); ``Rulebook.indice_pairs()`` gives spconv's (2,K,N)/(K,) view on demand.
"""
import os
import threading

import numpy as np
import torch

from .. import _lib
from .._lib import MODE_CONV, MODE_SUBM, MODE_TRANSPOSE, check, fast, i3, i3p, lib, ptr, stream_ptr, workspace


class LaunchProfile(object):
    """Optional per-launch instrumentation (bench.py's `roofline` leg): HIP events recorded on the stream the
    kernels are launched on, plus the ALGORITHMIC bytes / flops of each launch (SURVEY.md §8d formulas)."""

    def __init__(self):
        self.records = []  # (name, start_event, end_event, bytes, flops)

    def span(self, name, nbytes, flops, info=None):
        return _Span(self, name, nbytes, flops, info)

    def details(self):
        """per-launch rows: (name, ms, bytes, flops, info)"""
        return [(r[0], r[1].elapsed_time(r[2]), r[3], r[4], r[5] if len(r) > 5 else None) for r in self.records]

    def summary(self):
        out = {}
        for name, e0, e1, nbytes, flops in [r[:5] for r in self.records]:
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "bytes": 0, "flops": 0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["bytes"] += nbytes
            d["flops"] += flops
        return out


class _Span(object):
    def __init__(self, prof, name, nbytes, flops, info=None):
        self.prof, self.name, self.nbytes, self.flops, self.info = prof, name, nbytes, flops, info

    def __enter__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e1 = torch.cuda.Event(enable_timing=True)
        self.e0.record()

    def __exit__(self, *a):
        self.e1.record()
        self.prof.records.append((self.name, self.e0, self.e1, self.nbytes, self.flops, self.info))


class _NoSpan(object):
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


PROFILE = None  # set to a LaunchProfile() to instrument
CAPTURE = None  # set to a list to record (features, weight, bias, map_fwd, map_bwd) of every sparse conv (tools/conv_bench.py)
OVERLAP_WGRAD = True  # run wgrad on a side stream concurrently with dgrad (backward of every sparse conv)
_SIDE = {}


def _side_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _SIDE.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _SIDE[key] = st
    return st
_NOSPAN = _NoSpan()


def _span(name, cost_fn, args):
    """profile span around a launch; cost_fn(*args) is evaluated only while profiling (no closure on the hot path)"""
    if PROFILE is None:
        return _NOSPAN
    r = cost_fn(*args)
    return PROFILE.span(name, r[0], r[1], r[2] if len(r) > 2 else None)


def _num_pairs(nbr):
    """number of (in,out) pairs of a neighbour map = sum_k P_k (profile mode only; one sync per distinct map)"""
    n = getattr(nbr, "_btc_pairs", None)
    if n is None:
        n = int((nbr >= 0).sum().item())
        nbr._btc_pairs = n
    return n


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    """spconv.ops.get_conv_output_size: o = (i + 2p - d(k-1) - 1)//s + 1."""
    return [int((int(i) + 2 * p - d * (k - 1) - 1) // s + 1)
            for i, k, s, p, d in zip(input_size, kernel_size, stride, padding, dilation)]


def get_deconv_output_size(input_size, kernel_size, stride, padding, dilation, output_padding):
    """spconv.ops.get_deconv_output_size: o = (i-1)s - 2p + k + outpad."""
    return [int((int(i) - 1) * s - 2 * p + k + op)
            for i, k, s, p, op in zip(input_size, kernel_size, stride, padding, output_padding)]


class Rulebook(object):
    """One cached rulebook (what spconv stores in ``indice_dict[indice_key]``).

    A SUBMANIFOLD rulebook holds ONE map: its backward map is the forward map with the offset index mirrored
    (nbr_in[i][K-1-k] == nbr_out[i][k]), so ``map_bwd`` is the SAME tensor as ``nbr_out`` and the dgrad kernels read it mirrored
    (BTC_PASS_DGRAD_MIRROR, include/btcdet_hip.h) -- the bindings recognise that case by the identity of the two tensors.
    ``nbr_in`` still answers with the explicit (n, K) map (built on demand, cached), for tests and for spconv-style consumers."""

    __slots__ = ("out_indices", "in_indices", "nbr_out", "map_bwd", "_nbr_in", "in_shape", "out_shape", "K", "mode", "order_out", "order_in")

    def __init__(self, out_indices, in_indices, nbr_out, nbr_in, in_shape, out_shape, K, mode, order_out=None, order_in=None):
        self.out_indices, self.in_indices = out_indices, in_indices
        self.nbr_out = nbr_out
        mirrored = nbr_in is None or nbr_in is nbr_out or (mode == MODE_SUBM and nbr_out.numel() > 0 and nbr_in.data_ptr() == nbr_out.data_ptr())
        if mirrored and mode != MODE_SUBM:
            raise _lib.BtcHipError("only a submanifold rulebook can do without its backward map")
        self.map_bwd = nbr_out if mirrored else nbr_in      # what the kernels are handed
        self._nbr_in = None if mirrored else nbr_in
        self.in_shape, self.out_shape = list(in_shape), list(out_shape)
        self.K, self.mode = K, mode
        # row-order hints of the two maps (csrc/row_order.hip): int32 permutations the apply kernels tile the rows by, or
        # None = map order.  Built for strided / transposed rulebooks (their 16-row tiles are 17-30 % full in map order).
        # A TRANSPOSED layer's backward map is the opposite case: every input row reaches (nearly) all K offsets, there is nothing to
        # group -- and tiling it by first offset only scatters a tile's gathers over the whole 8x larger output level: the occupancy
        # net's deconv5 dgrad, 23 K rows gathering from 186 K, 31 us in map order and 178 us with the hint.  It keeps the map order.
        self.order_out, self.order_in = order_out, (None if mode == MODE_TRANSPOSE else order_in)

    @property
    def mirrored(self):
        return self.map_bwd is self.nbr_out

    @property
    def nbr_in(self):
        """the explicit backward map (n_in, K); for a submanifold rulebook the mirror image of nbr_out, materialised on first use"""
        if self._nbr_in is None:
            self._nbr_in = torch.flip(self.nbr_out, dims=[1]).contiguous()
        return self._nbr_in

    @property
    def n_in(self):
        return self.in_indices.shape[0]

    @property
    def n_out(self):
        return self.out_indices.shape[0]

    def indice_pairs(self):
        """spconv layout: indice_pairs (2,K,n_in) int32 padded with -1, indice_pair_num (K,) int32;
        pairs of an offset are ordered by output row (canonical order)."""
        dev = self.nbr_out.device
        pairs = torch.empty((2, self.K, max(self.n_in, 1)), dtype=torch.int32, device=dev)
        num = torch.empty((self.K,), dtype=torch.int32, device=dev)
        ws_bytes = lib().btc_pairs_from_nbr_ws_bytes(self.n_out, self.K)
        ws = workspace(ws_bytes, dev)
        check(lib().btc_pairs_from_nbr(ptr(self.nbr_out), self.n_out, self.K, self.n_in, ptr(pairs), ptr(num), ptr(ws), ws_bytes,
                                       stream_ptr()), "btc_pairs_from_nbr")
        return pairs[:, :, :self.n_in], num


ROW_ORDER = 1  # row-order hints: 1 = strided / transposed rulebooks, 0 = none (tests flip it)


def row_orders(maps):
    """row-order hints (csrc/row_order.hip) of several (n, K) neighbour maps from ONE launch: rows grouped by their first
    present offset (stable) inside blocks of 2048 rows; one int32 permutation per map (slices of one buffer).  Any permutation gives the same
    conv results."""
    import ctypes
    maps = list(maps)
    out = []
    for base in range(0, len(maps), 64):
        part = maps[base:base + 64]
        m = len(part)
        total = sum(int(t.shape[0]) for t in part)
        dev = part[0].device
        order = torch.empty((max(total, 1),), dtype=torch.int32, device=dev)
        ns = (ctypes.c_int32 * m)(*[int(t.shape[0]) for t in part])
        ks = (ctypes.c_int32 * m)(*[int(t.shape[1]) for t in part])
        ps = (ctypes.c_void_p * m)(*[ptr(t) for t in part])
        check(lib().btc_row_orders(ps, ns, ks, m, ptr(order), stream_ptr()), "btc_row_orders")
        off = 0
        for t in part:
            out.append(order[off:off + int(t.shape[0])])
            off += int(t.shape[0])
    return out


def _with_orders(rb):
    if ROW_ORDER and rb.K <= 64 and rb.order_out is None and (rb.mode != MODE_SUBM or ROW_ORDER >= 2):
        rb.order_out, rb.order_in = row_orders([rb.nbr_out, rb.map_bwd])   # (a mirrored map groups the same rows: columns only swap places)
        if rb.mode == MODE_TRANSPOSE:
            rb.order_in = None     # (Rulebook.__init__: a transposed layer's backward map is dense, the hint only scatters its gathers)
    return rb


def _as_idx(indices):
    if indices.dtype != torch.int32:
        indices = indices.int()
    return indices.contiguous()


class _Geometry(object):
    """host-side constants of one rulebook geometry, built once: int32[3] arrays, their ctypes pointers, K, mode, output
    shape (the launch-rate-bound forward builds ~18 rulebooks per step; re-deriving these cost ~25 us each)"""
    __slots__ = ("in_sh", "out_sh", "k3", "s3", "p3", "d3", "K", "mode", "subm", "p_in", "p_out", "p_k", "p_s", "p_p", "p_d",
                 "a_in", "a_out", "a_k", "a_s", "a_p", "a_d", "in_list", "out_list", "ws_bytes")

    def __init__(self, spatial_shape, ksize, stride, padding, dilation, out_padding, subm, transpose):
        self.k3, self.s3, self.p3, self.d3 = i3(ksize), i3(stride), i3(padding), i3(dilation)
        op3 = i3(out_padding)
        self.in_sh = i3([int(v) for v in spatial_shape])
        self.K = int(np.prod(self.k3))
        self.subm = bool(subm)
        self.mode = MODE_SUBM if subm else (MODE_TRANSPOSE if transpose else MODE_CONV)
        out_sh = np.zeros(3, dtype=np.int32)
        check(lib().btc_out_shape(i3p(self.in_sh), i3p(self.k3), i3p(self.s3), i3p(self.p3), i3p(self.d3), i3p(op3), self.mode,
                                  out_sh.ctypes.data_as(_lib.c_i32p)), "btc_out_shape")
        self.out_sh = i3(out_sh)
        self.p_in, self.p_out, self.p_k = i3p(self.in_sh), i3p(self.out_sh), i3p(self.k3)
        self.p_s, self.p_p, self.p_d = i3p(self.s3), i3p(self.p3), i3p(self.d3)
        self.in_list, self.out_list = self.in_sh.tolist(), self.out_sh.tolist()
        # raw addresses of the (interned, read-only) int32[3] arrays for the compiled binding
        self.a_in, self.a_out, self.a_k = self.in_sh.ctypes.data, self.out_sh.ctypes.data, self.k3.ctypes.data
        self.a_s, self.a_p, self.a_d = self.s3.ctypes.data, self.p3.ctypes.data, self.d3.ctypes.data
        self.ws_bytes = {}  # batch size -> conv workspace bytes


_GEOMETRIES = {}


def _geometry(spatial_shape, ksize, stride, padding, dilation, out_padding, subm, transpose):
    def t3(v):
        return (int(v),) * 3 if isinstance(v, (int, np.integer)) else tuple(int(x) for x in v)
    key = (tuple(int(v) for v in spatial_shape), t3(ksize), t3(stride), t3(padding), t3(dilation), t3(out_padding), bool(subm), bool(transpose))
    g = _GEOMETRIES.get(key)
    if g is None:
        g = _GEOMETRIES[key] = _Geometry(spatial_shape, ksize, stride, padding, dilation, out_padding, subm, transpose)
    return g


def build_rulebook(indices, batch_size, spatial_shape, ksize, stride=1, padding=0, dilation=1, out_padding=0,
                   subm=False, transpose=False):
    """ops.get_indice_pairs replacement.  indices (N,4) int32 [b,z,y,x] on the GPU."""
    indices = _as_idx(indices)
    if indices.dim() != 2 or indices.shape[1] != 4:
        raise _lib.BtcHipError(f"indices must be (N,4) [b,z,y,x], got {tuple(indices.shape)}")
    g = _geometry(spatial_shape, ksize, stride, padding, dilation, out_padding, subm, transpose)
    if PROFILE is not None and subm:
        return _build_rulebook_profiled(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, out_padding,
                                        subm, transpose)
    return _build_rulebook(indices, batch_size, g)


def build_rulebook_g(indices, batch_size, g):
    """build_rulebook for a prepared geometry (the layers keep theirs, spconv/conv.py)"""
    indices = _as_idx(indices)
    if PROFILE is not None and g.subm:
        return _build_rulebook_profiled(indices, batch_size, g.in_list, g.k3, g.s3, g.p3, g.d3, 0, True, False)
    return _build_rulebook(indices, batch_size, g)


def _build_rulebook_profiled(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, out_padding, subm, transpose):
    global PROFILE
    prof, PROFILE = PROFILE, None
    try:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rb = build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, out_padding, subm, transpose)
        e1.record()
        # SURVEY.md §8d, rulebook: 16 N_in + 16 N_out + 8 sum_k P_k bytes
        prof.records.append(("rulebook", e0, e1, 16 * rb.n_in + 16 * rb.n_out + 8 * _num_pairs(rb.nbr_out), 0,
                             dict(rows=rb.n_out, n_in=rb.n_in, K=rb.K, pairs=_num_pairs(rb.nbr_out), mode=rb.mode,
                                  out_shape=list(rb.out_shape))))
    finally:
        PROFILE = prof
    return rb


def _build_rulebook(indices, batch_size, g):
    dev = indices.device
    n = indices.shape[0]
    L = lib()
    K = g.K
    F = fast() if PROFILE is None else None
    if F is not None:
        if g.subm:
            nbr = F.rulebook_subm(indices, int(batch_size), g.a_in, g.a_k, g.a_d, K, stream_ptr())
            return Rulebook(indices, indices, nbr, None, g.in_list, g.out_list, K, g.mode)
        out_indices, nbr_out, nbr_in, o_out, o_in = F.rulebook_conv(indices, int(batch_size), g.a_in, g.a_out, g.a_k, g.a_s, g.a_p, g.a_d, g.mode, K,
                                                                    _conv_ws_bytes(g, batch_size), stream_ptr())
        if not ROW_ORDER:
            o_out = o_in = None
        return Rulebook(out_indices, indices, nbr_out, nbr_in, g.in_list, g.out_list, K, g.mode, o_out, o_in)
    if g.subm:
        nbr_out = torch.empty((n, K), dtype=torch.int32, device=dev)   # nbr_in is its mirror image: not built (Rulebook docstring)
        ws_bytes = L.btc_rulebook_subm_ws_bytes(n)
        ws = workspace(ws_bytes, dev)
        check(L.btc_rulebook_subm(ptr(indices), n, int(batch_size), g.p_in, g.p_k, g.p_d, ptr(nbr_out), None, ptr(ws), ws_bytes,
                                  stream_ptr()), "btc_rulebook_subm")
        return Rulebook(indices, indices, nbr_out, None, g.in_list, g.out_list, K, g.mode)
    if PROFILE is not None:
        return _start_conv_rulebook(indices, batch_size, g, None).finish()
    # synchronous build: count, one blocking 4-byte read-back (spconv syncs at the same point), fill
    ws_bytes = _conv_ws_bytes(g, batch_size)
    ws = workspace(ws_bytes, dev)
    d_n_out = torch.empty((1,), dtype=torch.int32, device=dev)
    st = stream_ptr()
    check(L.btc_rulebook_conv_count(ptr(indices), n, int(batch_size), g.p_in, g.p_out, g.p_k, g.p_s, g.p_p, g.p_d, g.mode, ptr(d_n_out),
                                    ptr(ws), ws_bytes, st), "btc_rulebook_conv_count")
    return _fill_conv_rulebook(indices, batch_size, g, int(d_n_out.item()), ws, ws_bytes)


def _conv_ws_bytes(g, batch_size):
    b = g.ws_bytes.get(int(batch_size))
    if b is None:
        b = g.ws_bytes[int(batch_size)] = lib().btc_rulebook_conv_ws_bytes(int(batch_size), g.p_out)
    return b


def _fill_conv_rulebook(indices, batch_size, g, n_out, ws, ws_bytes):
    dev, n, K = indices.device, indices.shape[0], g.K
    out_indices = torch.empty((n_out, 4), dtype=torch.int32, device=dev)
    nbr_out = torch.empty((n_out, K), dtype=torch.int32, device=dev)
    nbr_in = torch.empty((n, K), dtype=torch.int32, device=dev)
    check(lib().btc_rulebook_conv_fill(ptr(indices), n, int(batch_size), g.p_in, g.p_out, g.p_k, g.p_s, g.p_p, g.p_d, g.mode, n_out,
                                       ptr(out_indices), ptr(nbr_out), ptr(nbr_in), ptr(ws), ws_bytes, stream_ptr()), "btc_rulebook_conv_fill")
    return _with_orders(Rulebook(out_indices, indices, nbr_out, nbr_in, g.in_list, g.out_list, K, g.mode))


# ---- strided / transposed rulebooks in two halves -------------------------------------------------------------------
# The number of output rows of such a rulebook is data dependent, and every tensor downstream is sized by it: the host
# must read it back (spconv syncs at the same point).  A blocking read-back drains the stream -- and after each one the
# host has to refill the queue kernel by kernel while the GPU runs dry (13 read-backs per step made ~2 of 10.6 ms idle).
# So the COUNT half (mark reachable cells, rank them, n_out -> pinned host memory) runs on a side stream as soon as the
# input indices exist -- a layer's `lookahead` list names the strided layers that consume its output level
# (spconv/conv.py) -- and the FILL half runs in the consumer's forward: by then the count is long finished, the host
# waits on its event only (not on the main stream) and keeps running ahead of the GPU.
# Measured at KITTI size: with the halves managed in Python (ctypes route) neutral to negative -- stream contexts, events
# and the pinned copy cost the host what the wait saved; managed inside the compiled binding (_btcfast.rulebook_conv_start /
# _finish) +1.5 % (221 -> 224.5 scenes/s) -- see lookahead_enabled().
def lookahead_enabled():
    """on when the compiled binding manages the two halves (events, pinned slot and side stream in C++: 221 -> 224.5 scenes/s), off on
    the ctypes route (its Python per event costs what the wait saved)"""
    return fast() is not None


_RB_STREAM = {}
_PIN = {}


def _rb_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _RB_STREAM.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _RB_STREAM[key] = st
    return st


_PIN_LOCK = threading.Lock()


def _pinned_slot():
    """one int32 of pinned host memory from a small ring (a slot is reused 256 read-backs later); prepare() may run on a worker
    thread beside the training thread, so slots are handed out under a lock"""
    with _PIN_LOCK:
        ring = _PIN.get("ring")
        if ring is None:
            ring = _PIN["ring"] = torch.zeros((256,), dtype=torch.int32).pin_memory()
            _PIN["next"] = 0
        i = _PIN["next"]
        _PIN["next"] = (i + 1) % 256
    return ring[i:i + 1]


class PendingRulebook(object):
    """count half issued, fill half outstanding"""

    def __init__(self, indices, batch_size, g, ws, ws_bytes, host_n, event, prof_ev):
        self.indices, self.batch_size, self.g = indices, batch_size, g
        self.ws, self.ws_bytes, self.host_n, self.event, self.prof_ev = ws, ws_bytes, host_n, event, prof_ev

    def finish(self):
        self.event.synchronize()                        # host: the count (on the side stream) is done; the main stream is not drained
        n_out = int(self.host_n[0])
        main = torch.cuda.current_stream()
        main.wait_event(self.event)                     # device: the fill below reads the bitmap / ranks the count wrote
        self.ws.record_stream(main)
        prof = PROFILE
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rb = _fill_conv_rulebook(self.indices, self.batch_size, self.g, n_out, self.ws, self.ws_bytes)
        if prof is not None:
            e1.record()
            c0, c1 = self.prof_ev
            # SURVEY.md §8d, rulebook: 16 N_in + 16 N_out + 8 sum_k P_k bytes (attributed to the fill half; the count half adds time only)
            info = dict(rows=rb.n_out, n_in=rb.n_in, K=rb.K, pairs=_num_pairs(rb.nbr_out), mode=rb.mode, out_shape=list(rb.out_shape))
            prof.records.append(("rulebook", c0, c1, 0, 0, dict(info, half="count")))
            prof.records.append(("rulebook", e0, e1, 16 * rb.n_in + 16 * rb.n_out + 8 * _num_pairs(rb.nbr_out), 0, dict(info, half="fill")))
        self.ws = None
        return rb


def _start_conv_rulebook(indices, batch_size, g, side):
    """issue the count half on `side` (a torch stream) or, side=None, on the current stream"""
    dev, n, L = indices.device, indices.shape[0], lib()
    ws_bytes = _conv_ws_bytes(g, batch_size)
    if side is not None:
        side.wait_stream(torch.cuda.current_stream())   # the indices are produced on the main stream
    ctx = torch.cuda.stream(side) if side is not None else _NOSPAN
    prof_ev = None
    with ctx:
        if PROFILE is not None:
            prof_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            prof_ev[0].record()
        ws = workspace(ws_bytes, dev)
        d_n_out = torch.empty((1,), dtype=torch.int32, device=dev)
        check(L.btc_rulebook_conv_count(ptr(indices), n, int(batch_size), g.p_in, g.p_out, g.p_k, g.p_s, g.p_p, g.p_d, g.mode, ptr(d_n_out),
                                        ptr(ws), ws_bytes, stream_ptr()), "btc_rulebook_conv_count")
        if prof_ev is not None:
            prof_ev[1].record()
        host_n = _pinned_slot()
        host_n.copy_(d_n_out, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
    if side is not None:
        indices.record_stream(side)
    return PendingRulebook(indices, batch_size, g, ws, ws_bytes, host_n, event, prof_ev)


class _NativePending(PendingRulebook):
    """the same two halves managed inside the compiled binding (events, pinned read-back slot and side stream in C++)"""

    def __init__(self, F, handle, indices, g):
        self.F, self.handle, self.indices, self.g = F, handle, indices, g

    def finish(self):
        g = self.g
        out_indices, nbr_out, nbr_in, o_out, o_in = self.F.rulebook_conv_finish(self.handle)
        self.handle = None
        if not ROW_ORDER:
            o_out = o_in = None
        return Rulebook(out_indices, self.indices, nbr_out, nbr_in, g.in_list, g.out_list, g.K, g.mode, o_out, o_in)


def prefetch_conv_rulebook(indices, batch_size, spatial_shape, ksize, stride=1, padding=0, dilation=1, out_padding=0, transpose=False):
    """count half of a strided / transposed rulebook on the side stream; .finish() on the result gives the Rulebook"""
    indices = _as_idx(indices)
    g = _geometry(spatial_shape, ksize, stride, padding, dilation, out_padding, False, transpose)
    F = fast() if PROFILE is None else None
    if F is not None:
        h = F.rulebook_conv_start(indices, int(batch_size), g.a_in, g.a_out, g.a_k, g.a_s, g.a_p, g.a_d, g.mode, g.K, _conv_ws_bytes(g, batch_size))
        return _NativePending(F, h, indices, g)
    return _start_conv_rulebook(indices, batch_size, g, _rb_stream(indices.device))


def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1, out_padding=0,
                     subm=False, transpose=False, grid=None):
    """spconv.ops.get_indice_pairs signature -> (outids, indice_pairs, indice_pair_num)."""
    rb = build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, out_padding, subm, transpose)
    pairs, num = rb.indice_pairs()
    return rb.out_indices, pairs, num


def _f32c(t):
    if t.dtype != torch.float32:
        raise _lib.BtcHipError(f"float32 features expected, got {t.dtype}")
    return t.contiguous()


def _actc(t):
    """activations: float32, or bfloat16 ("bf16 features", BASELINE.json configs[2])"""
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.BtcHipError(f"float32 or bfloat16 features expected, got {t.dtype}")
    return t.contiguous()


def _conv_cost(nbr, n_res, K, cred, cres, act_bytes=4):
    """SURVEY.md §8d, sparse conv fwd (dgrad is the same launch with the channel roles swapped):
    bytes = s (sum_k P_k (Cin + Cout) + N_out Cout) + 4 K Cin Cout, flops = 2 sum_k P_k Cin Cout (s = 4 fp32 / 2 bf16 activations)"""
    P = _num_pairs(nbr)
    return (act_bytes * (P * (cred + cres) + n_res * cres) + 4 * K * cred * cres, 2 * P * cred * cres,
            dict(rows=n_res, K=K, cred=cred, cres=cres, pairs=P))


def _wgrad_cost(nbr, n_res, K, cin, cout, act_bytes=4):
    P = _num_pairs(nbr)
    return act_bytes * P * (cin + cout) + 4 * K * cin * cout, 2 * P * cin * cout, dict(rows=n_res, K=K, cred=cin, cres=cout, pairs=P)


# wgrad goes to a side HIP stream (fork / join with events, no host sync) beside dgrad for layers with at least this many
# rows; on smaller layers the stream switches cost more host time than the overlap returns (the forward pass is bound by
# the host's launch rate, the backward pass by the GPU: tools/host_phases.py).  Measured: 60000 -> 191, 20000 -> 194,
# 0 -> 194 scenes/s.  BTC_OVERLAP_MIN_ROWS in the environment (measurement scripts: a huge value = no kernel beside another).
NATIVE_AUTOGRAD = True  # conv -> BN -> ReLU as a C++ autograd node when _btcfast is built
OVERLAP_MIN_ROWS = int(os.environ.get("BTC_OVERLAP_MIN_ROWS", "20000"))
OVERLAP_MAX_ROWS = 100000  # above: both kernels fill the GPU alone, side by side 450 us vs 219 + 150


def set_defer_wgrad_join(on):
    """compiled binding only: run every weight gradient on the side stream and join it once, at the end of backward (an
    autograd-engine callback) -- weight gradients are off the critical path of the backward chain.  A caller that reads
    dW before backward returns (a gradient reducer launched from a hook) must call join_wgrad() first; DistributedDataParallel
    does read in mid-backward, so leave this off under DDP."""
    F = fast()
    if F is not None:
        F.set_defer_wgrad_join(bool(on))
    _defer_state[0] = bool(on) and F is not None
    return F is not None


_defer_state = [False]


def defer_wgrad_join_enabled():
    return _defer_state[0]


def join_wgrad():
    F = fast()
    if F is not None:
        F.join_wgrad()


def _overlap_ok(n_res):
    return OVERLAP_WGRAD and OVERLAP_MIN_ROWS <= n_res < OVERLAP_MAX_ROWS


_WQ = {}


def _bf16_operands(features, K, cred, cres):
    """bf16 activations + a bf16 copy of the weights on the bf16 matrix pipe (csrc/conv_apply_bf16.hip) unless the tuning key
    BTC_TUNE_BF16_OPERANDS is 1; same policy as the compiled binding (binding.cpp bf16_operands)"""
    L = lib()
    return features.dtype == torch.bfloat16 and L.btc_conv_bf16w_supported(int(K), int(cred), int(cres)) == 1 and L.btc_tune_value(8) != 1


def _weights_bf16(w, K, cin, cout):
    """(2, numel) bf16: row 0 = W [K][Cin][Cout], row 1 = W^T [K][Cout][Cin]; rebuilt when the parameter's version counter moved"""
    import weakref
    hit = _WQ.get(id(w)) if w.is_leaf else None
    if hit is not None and hit[0]() is w and hit[1] == w._version:
        return hit[2]
    q = torch.empty((2, w.numel()), dtype=torch.bfloat16, device=w.device)
    check(lib().btc_weights_to_bf16(ptr(w), int(K), int(cin), int(cout), ptr(q[0]), ptr(q[1]), stream_ptr()), "btc_weights_to_bf16")
    if not w.is_leaf:   # a temporary (the zero-padded 34 -> 48 channel weight is a fresh tensor every step): not cached
        return q
    for k in [k for k, v in _WQ.items() if v[0]() is None]:   # every miss drops the entries of freed parameters
        del _WQ[k]
    _WQ[id(w)] = (weakref.ref(w), w._version, q)
    return q


_WS3 = {}
_SCRATCH = {}


def _split_operands(src, K, cred, cres, n_rows):
    """fp32 launch on the split-operand kernel (csrc/conv_apply_split.hip: three bf16 pieces per operand, six bf16 MFMAs per
    product block)?  The library's policy (btc_conv_split_wanted; never under BTC_TUNE_SPLIT = 1) -- same as binding.cpp"""
    if not (src.dtype == torch.float32 and lib().btc_conv_split_wanted(int(K), int(cred), int(cres), int(n_rows)) == 1):
        return False
    if src.numel() * 4 >= 0xFFFFFF00:    # the kernel's gathers use 32-bit byte offsets (it traps past them): the exact kernels take any size
        return False
    key = (src.device.index, stream_ptr())
    if key not in _SCRATCH:    # the stream's scratch buffer for z-split launches (btc_set_scratch), as binding.cpp ensure_scratch
        buf = _SCRATCH[key] = torch.empty(48 << 20, dtype=torch.uint8, device=src.device)
        check(lib().btc_set_scratch(stream_ptr(), ptr(buf), buf.numel()), "btc_set_scratch")
    return True


def _weights_split(w, K, cin, cout):
    """(2, 3 * numel) bf16: row 0 = the hi | mid | lo planes of W [K][Cin][Cout] (dgrad operand), row 1 = those of W^T (forward
    operand); rebuilt when the parameter's version counter moved"""
    import weakref
    hit = _WS3.get(id(w)) if w.is_leaf else None
    if hit is not None and hit[0]() is w and hit[1] == w._version:
        return hit[2]
    q = torch.empty((2, 3 * w.numel()), dtype=torch.bfloat16, device=w.device)
    check(lib().btc_weights_split3(ptr(w), int(K), int(cin), int(cout), ptr(q[0]), ptr(q[1]), stream_ptr()), "btc_weights_split3")
    if not w.is_leaf:
        return q
    for k in [k for k, v in _WS3.items() if v[0]() is None]:
        del _WS3[k]
    _WS3[id(w)] = (weakref.ref(w), w._version, q)
    return q


def _conv_forward(features, w, b, map_fwd, ord_fwd=None):
    if PROFILE is None:
        F = fast()
        if F is not None:
            return F.conv_fwd(features, w, b, map_fwd, ord_fwd, stream_ptr())
    bf = features.dtype == torch.bfloat16
    cin, cout = w.shape[-2], w.shape[-1]
    K = map_fwd.shape[1]
    if w.numel() != K * cin * cout or features.shape[1] != cin:
        raise _lib.BtcHipError(f"weight {tuple(w.shape)} does not match K={K}, Cin={features.shape[1]}")
    n_res = map_fwd.shape[0]
    out = torch.empty((n_res, cout), dtype=features.dtype, device=features.device)
    if _bf16_operands(features, K, cin, cout):
        q = _weights_bf16(w, K, cin, cout)
        with _span("conv_apply", _conv_cost, (map_fwd, n_res, K, cin, cout, 2)):
            check(lib().btc_conv_apply_ordered(0, 2, ptr(features), ptr(q[1]), ptr(b), ptr(map_fwd), ptr(ord_fwd), n_res, K, cin, cout, ptr(out),
                                               stream_ptr()), "btc_conv_apply_ordered")
        return out
    if _split_operands(features, K, cin, cout, n_res):
        q = _weights_split(w, K, cin, cout)
        with _span("conv_apply", _conv_cost, (map_fwd, n_res, K, cin, cout, 4)):
            check(lib().btc_conv_apply_src(0, 3, ptr(features), int(features.shape[0]), ptr(q[1]), ptr(b), ptr(map_fwd), ptr(ord_fwd), n_res, K, cin, cout,
                                           ptr(out), stream_ptr()), "btc_conv_apply_src")
        return out
    with _span("conv_apply", _conv_cost, (map_fwd, n_res, K, cin, cout, 2 if bf else 4)):
        check(lib().btc_conv_apply_ordered(0, 1 if bf else 0, ptr(features), ptr(w), ptr(b), ptr(map_fwd), ptr(ord_fwd), n_res, K, cin, cout,
                                           ptr(out), stream_ptr()), "btc_conv_apply_ordered")
    return out


def _conv_backward(features, w, map_fwd, map_bwd, grad_out, wshape, need_din, need_dw, allow_defer=False, ord_bwd=None):
    bf = features.dtype == torch.bfloat16
    cin, cout = w.shape[-2], w.shape[-1]
    K = map_fwd.shape[1]
    L = lib()
    wgrad = L.btc_conv_wgrad_bf16 if bf else L.btc_conv_wgrad
    din = dw = None
    dev = grad_out.device
    n_res, n_src = map_fwd.shape[0], map_bwd.shape[0]
    mirror = map_bwd is map_fwd or (map_fwd.numel() > 0 and map_bwd.data_ptr() == map_fwd.data_ptr())   # a submanifold rulebook's single map (Rulebook docstring)
    pass_dgrad = 2 if mirror else 1                                            # BTC_PASS_DGRAD_MIRROR / BTC_PASS_DGRAD
    wg_bwd = map_bwd   # (weight gradient: the same pointer twice says "one map, mirrored"; btcdet_hip.h)
    overlap = bool(need_din and need_dw and _overlap_ok(n_res))
    if PROFILE is None:
        F = fast()
        if F is not None:
            return F.conv_bwd(features, w, map_fwd, map_bwd, ord_bwd, grad_out, bool(need_din), bool(need_dw), overlap, bool(allow_defer), stream_ptr())
    side = _side_stream(dev) if (overlap and PROFILE is None) else None
    if need_dw:
        ws_bytes = L.btc_conv_wgrad_ws_bytes(n_res, K, cin, cout, n_src)
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                dw = torch.empty(wshape, dtype=torch.float32, device=dev)
                ws = workspace(ws_bytes, dev)
                check(wgrad(ptr(features), ptr(grad_out), ptr(map_fwd), n_res, ptr(wg_bwd), n_src, K, cin, cout, ptr(dw), ptr(ws), ws_bytes,
                            stream_ptr()), "btc_conv_wgrad")
            for t in (features, grad_out, map_fwd, map_bwd):
                t.record_stream(side)
        else:
            dw = torch.empty(wshape, dtype=torch.float32, device=dev)
            ws = workspace(ws_bytes, dev)
            with _span("conv_wgrad", _wgrad_cost, (map_fwd, n_res, K, cin, cout, 2 if bf else 4)):
                check(wgrad(ptr(features), ptr(grad_out), ptr(map_fwd), n_res, ptr(wg_bwd), n_src, K, cin, cout, ptr(dw), ptr(ws), ws_bytes,
                            stream_ptr()), "btc_conv_wgrad")
    if need_din:
        din = torch.empty((n_src, cin), dtype=features.dtype, device=dev)
        with _span("conv_apply", _conv_cost, (map_bwd, n_src, K, cout, cin, 2 if bf else 4)):
            if _bf16_operands(grad_out, K, cout, cin):
                q = _weights_bf16(w, K, cin, cout)
                check(L.btc_conv_apply_ordered(pass_dgrad, 2, ptr(grad_out), ptr(q[0]), None, ptr(map_bwd), ptr(ord_bwd), n_src, K, cin, cout, ptr(din),
                                               stream_ptr()), "btc_conv_apply_ordered")
            elif _split_operands(grad_out, K, cout, cin, n_src):
                q = _weights_split(w, K, cin, cout)
                check(L.btc_conv_apply_src(pass_dgrad, 3, ptr(grad_out), int(grad_out.shape[0]), ptr(q[0]), None, ptr(map_bwd), ptr(ord_bwd), n_src, K, cin,
                                           cout, ptr(din), stream_ptr()), "btc_conv_apply_src")
            else:
                check(L.btc_conv_apply_ordered(pass_dgrad, 1 if bf else 0, ptr(grad_out), ptr(w), None, ptr(map_bwd), ptr(ord_bwd), n_src, K, cin, cout,
                                               ptr(din), stream_ptr()), "btc_conv_apply_ordered")
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)  # join: dW is consumed on the main stream from here on
        dw.record_stream(torch.cuda.current_stream())
    return din, dw


def _bias_grad(g):
    if g.is_cuda and g.dim() == 2 and g.shape[0] >= 1 and g.is_contiguous() and g.dtype in (torch.float32, torch.bfloat16):
        from . import fused_bn
        return fused_bn.col_sum(g)
    return g.sum(0, dtype=torch.float32)


class SparseConvFunction(torch.autograd.Function):
    """indice_conv / indice_subm_conv / indice_inverse_conv in one function.
    map_fwd (n_res,K): source row gathered by result row i at offset k; map_bwd (n_src,K) its transpose."""

    @staticmethod
    def forward(ctx, features, weight, bias, map_fwd, map_bwd, ord_fwd=None, ord_bwd=None):
        features = _actc(features)
        w = _f32c(weight)
        b = _f32c(bias) if bias is not None else None
        out = _conv_forward(features, w, b, map_fwd, ord_fwd)
        ctx.ord_bwd = ord_bwd
        if CAPTURE is not None:
            CAPTURE.append((features, w, b, map_fwd, map_bwd))
        ctx.save_for_backward(features, w, map_fwd, map_bwd)
        ctx.has_bias = bias is not None
        ctx.wshape = tuple(weight.shape)
        # a dW consumed by another autograd node (cat of head weights) must not be deferred unless that consumer joins first
        ctx.leaf_w = bool(weight.is_leaf or getattr(weight, "_btc_join_before_use", False))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        features, w, map_fwd, map_bwd = ctx.saved_tensors
        grad_out = _actc(grad_out if grad_out.dtype == features.dtype else grad_out.to(features.dtype))
        din, dw = _conv_backward(features, w, map_fwd, map_bwd, grad_out, ctx.wshape, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                 ctx.leaf_w, ctx.ord_bwd)
        db = _bias_grad(grad_out) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return din, dw, db, None, None, None, None


class SparseConvBNReLUFunction(torch.autograd.Function):
    """sparse conv -> BatchNorm1d (-> ReLU) as ONE autograd node (the reference's post_act_block triple,
    spconv_backbone.py:33-43): same kernels as SparseConvFunction + fused_bn.BatchNormReLUFunction, half the Python / autograd
    dispatch per layer -- the step is bound by the host's launch rate at BtcDet's sizes."""

    @staticmethod
    def forward(ctx, features, weight, bias, map_fwd, map_bwd, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, relu,
                ord_fwd=None, ord_bwd=None):
        from . import fused_bn
        features = _actc(features)
        w = _f32c(weight)
        b = _f32c(bias) if bias is not None else None
        use_batch = bool(training or running_mean is None)
        F = fast() if PROFILE is None else None
        if F is not None:
            ws, need = fused_bn._ws(features.device, w.shape[-1])
            x, y, stats = F.conv_bn_fwd(features, w, b, map_fwd, ord_fwd, gamma, beta, running_mean, running_var, nbt if training else None, use_batch,
                                        float(momentum), float(eps), bool(relu), ws, need, stream_ptr())
        elif (use_batch and PROFILE is None and map_fwd.shape[0] >= 1 and w.numel() == map_fwd.shape[1] * w.shape[-2] * w.shape[-1]
              and features.shape[1] == w.shape[-2]):
            # the ctypes route of the same fused entry point the compiled binding takes (statistics in the conv's epilogue)
            x, y, stats = fused_bn.conv_bn_forward(features, w, b, map_fwd, ord_fwd, gamma, beta, running_mean, running_var,
                                                   nbt if training else None, momentum, eps, relu)
        else:
            x = _conv_forward(features, w, b, map_fwd, ord_fwd)
            y, stats = fused_bn.bn_forward(x, gamma, beta, running_mean, running_var, nbt if training else None, use_batch, momentum, eps, relu)
        if CAPTURE is not None:
            CAPTURE.append((features, w, b, map_fwd, map_bwd))
        ctx.save_for_backward(features, w, map_fwd, map_bwd, x, y, gamma, stats)
        ctx.flags = (bias is not None, tuple(weight.shape), use_batch, bool(relu), bool(weight.is_leaf))
        ctx.ord_bwd = ord_bwd
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import fused_bn
        features, w, map_fwd, map_bwd, x, y, gamma, stats = ctx.saved_tensors
        has_bias, wshape, use_batch, relu, leaf_w = ctx.flags
        dy = (dy if dy.dtype == x.dtype else dy.to(x.dtype)).contiguous()
        dx, dgamma, dbeta = fused_bn.bn_backward(x, y, dy, gamma, stats, use_batch, relu)
        din, dw = _conv_backward(features, w, map_fwd, map_bwd, dx, wshape, ctx.needs_input_grad[0], ctx.needs_input_grad[1], leaf_w, ctx.ord_bwd)
        db = _bias_grad(dx) if (has_bias and ctx.needs_input_grad[2]) else None
        affine = gamma is not None
        return (din, dw, db, None, None, dgamma if affine else None, dbeta if affine else None, None, None, None, None, None, None, None, None,
                None)


class SparseMaxPoolFunction(torch.autograd.Function):
    """indice_maxpool."""

    @staticmethod
    def forward(ctx, features, nbr_out, nbr_in):
        features = _f32c(features)
        n_out, K = nbr_out.shape
        C = features.shape[1]
        out = torch.empty((n_out, C), dtype=torch.float32, device=features.device)
        check(lib().btc_maxpool_fwd(ptr(features), ptr(nbr_out), n_out, K, C, ptr(out), stream_ptr()), "btc_maxpool_fwd")
        ctx.save_for_backward(features, out, nbr_in)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        features, out, nbr_in = ctx.saved_tensors
        grad_out = _f32c(grad_out)
        n_in, K = nbr_in.shape
        C = features.shape[1]
        din = torch.empty_like(features)
        check(lib().btc_maxpool_bwd(ptr(features), ptr(out), ptr(grad_out), ptr(nbr_in), n_in, K, C, ptr(din),
                                    stream_ptr()), "btc_maxpool_bwd")
        return din, None, None


class ToDenseFunction(torch.autograd.Function):
    """SparseConvTensor.dense(): (N,C) rows -> zeros(B,C,D,H,W)."""

    @staticmethod
    def forward(ctx, features, indices, batch_size, spatial_shape):
        features = _f32c(features)
        indices = _as_idx(indices)
        n, C = features.shape
        sh = i3([int(v) for v in spatial_shape])
        dense = torch.zeros((int(batch_size), C, int(sh[0]), int(sh[1]), int(sh[2])), dtype=torch.float32,
                            device=features.device)
        check(lib().btc_dense_fwd(ptr(features), ptr(indices), n, C, i3p(sh), ptr(dense), stream_ptr()), "btc_dense_fwd")
        ctx.save_for_backward(indices)
        ctx.sh = sh
        ctx.nc = (n, C)
        return dense

    @staticmethod
    def backward(ctx, grad_dense):
        (indices,) = ctx.saved_tensors
        n, C = ctx.nc
        grad_dense = _f32c(grad_dense)
        dfeat = torch.empty((n, C), dtype=torch.float32, device=grad_dense.device)
        check(lib().btc_dense_bwd(ptr(grad_dense), ptr(indices), n, C, i3p(ctx.sh), ptr(dfeat), stream_ptr()),
              "btc_dense_bwd")
        return dfeat, None, None, None


class DenseSplitFunction(torch.autograd.Function):
    """dense() of a tensor whose channels are two heads side by side (the occupancy head's merged conv_cls | conv_res output) into the
    two dense maps: one fill + one scatter launch forward, one gather launch backward (csrc/glue.hip) -- the same values as dense() of the
    two column slices, which cost two copies, two fills and two scatters (and four fills / copies + an add in backward)."""

    @staticmethod
    def forward(ctx, features, indices, batch_size, spatial_shape, ca):
        features = _f32c(features)
        indices = _as_idx(indices)
        n, C = features.shape
        cb = C - int(ca)
        sh = i3([int(v) for v in spatial_shape])
        B, vol = int(batch_size), int(sh[0]) * int(sh[1]) * int(sh[2])
        flat = torch.zeros((B * C * vol,), dtype=torch.float32, device=features.device)     # ONE fill for both maps
        da = flat[:B * ca * vol].view(B, ca, int(sh[0]), int(sh[1]), int(sh[2]))
        db = flat[B * ca * vol:].view(B, cb, int(sh[0]), int(sh[1]), int(sh[2]))
        check(lib().btc_dense_split_fwd(ptr(features), ptr(indices), n, int(ca), cb, i3p(sh), ptr(da), ptr(db), stream_ptr()), "btc_dense_split_fwd")
        ctx.save_for_backward(indices)
        ctx.meta = (sh, n, int(ca), cb)
        return da, db

    @staticmethod
    def backward(ctx, ga, gb):
        (indices,) = ctx.saved_tensors
        sh, n, ca, cb = ctx.meta
        ga = _f32c(ga) if ga is not None else None
        gb = _f32c(gb) if gb is not None else None
        dev = (ga if ga is not None else gb).device
        dfeat = torch.empty((n, ca + cb), dtype=torch.float32, device=dev)
        check(lib().btc_dense_split_bwd(ptr(ga), ptr(gb), ptr(indices), n, ca, cb, i3p(sh), ptr(dfeat), stream_ptr()), "btc_dense_split_bwd")
        return dfeat, None, None, None, None


def dense_split(features, indices, batch_size, spatial_shape, ca):
    """-> (dense of features[:, :ca], dense of features[:, ca:]), each (B, c, D, H, W) contiguous"""
    return DenseSplitFunction.apply(features, indices, batch_size, spatial_shape, ca)


class CatPadFunction(torch.autograd.Function):
    """[a | b | zeros] along the channels in one launch each way (csrc/glue.hip); fp32 or bf16 GPU tensors of one dtype"""

    @staticmethod
    def forward(ctx, a, b, cout):
        a, b = a.contiguous(), b.contiguous()
        n, ca, cb = a.shape[0], a.shape[1], b.shape[1]
        out = torch.empty((n, int(cout)), dtype=a.dtype, device=a.device)
        check(lib().btc_cat_pad_fwd(ptr(a), ca, ptr(b), cb, n, int(cout), a.element_size(), ptr(out), stream_ptr()), "btc_cat_pad_fwd")
        ctx.meta = (n, ca, cb, int(cout), a.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        n, ca, cb, cout, dt = ctx.meta
        g = (g if g.dtype == dt else g.to(dt)).contiguous()
        da = torch.empty((n, ca), dtype=dt, device=g.device)
        db = torch.empty((n, cb), dtype=dt, device=g.device)
        check(lib().btc_cat_pad_bwd(ptr(g), cout, n, g.element_size(), ptr(da), ca, ptr(db), cb, stream_ptr()), "btc_cat_pad_bwd")
        return da, db, None


def cat_features(a, b):
    """torch.cat((a, b), dim=1) of two (n, c) feature matrices.  When the layer behind it would zero-pad the sum's channel count
    (_pad_in_channels: 32 + 2 -> 64), the padding is appended HERE, in the same launch: the consumer finds its input already as wide as
    it wants it and only pads its weight."""
    b = b.to(a.dtype)
    cc = a.shape[1] + b.shape[1]
    if a.is_cuda and a.dtype in (torch.float32, torch.bfloat16) and a.dim() == 2 and a.shape[0] > 0 and pads_in_channels(cc):
        return CatPadFunction.apply(a, b, cc + _pad_amount(cc, a.dtype))
    return torch.cat((a, b), dim=1)


# fp32 features: 34 -> 64 channels (one 64-channel item per offset instead of three 16-channel ones: 83 -> 60 us forward, 68 -> 55 us dgrad
# at 29 K rows, and the dgrad becomes a 32 -> 64 layer the split-operand kernel takes); bf16 features: 34 -> 48, which keeps that layer on
# fp32 weights (48 is not a multiple of 32: no bf16 weight copy), i.e. bit-equal to the oracle's chain on the widened activations
PAD_CHANNELS = 0   # 0 = by dtype as above


def pads_in_channels(cin):
    """whether indice_conv zero-pads a layer's input channels (to the next multiple of PAD_CHANNELS)"""
    return cin > 16 and cin % 16 != 0


def _pad_amount(cin, dtype):
    return (-cin) % (PAD_CHANNELS or (32 if dtype == torch.float32 else 16))


def _pad_in_channels(features, weight):
    """A layer with 16 < Cin and Cin % 16 != 0 (the detection backbone's 34 -> 32 conv2_combine: 32 feature + 2 occupancy
    channels) misses the LDS-DMA apply kernel and the pipelined weight-gradient kernel by a few channels: measured 103 / 95 /
    176 us forward / dgrad / wgrad against 46 / 47 / 59 for the 32 -> 32 layer beside it.  Zero channels appended to the
    features and zero rows to the weight add exact zeros to every fmaf chain, so the result bits do not change; autograd slices
    the padding off the gradients."""
    cin = weight.shape[-2]
    if features.is_cuda and pads_in_channels(cin):
        # to the next multiple of PAD_CHANNELS: with 48 channels the LDS-DMA kernel walks 16-channel items (3 per offset, 81 per
        # workgroup for a 3 x 3 x 3 kernel), with 64 one 64-channel item per offset
        pad = _pad_amount(cin, features.dtype)
        if features.shape[1] == cin + pad:     # cat_features appended the zero channels already
            return features, torch.nn.functional.pad(weight, (0, 0, 0, pad))
        return torch.nn.functional.pad(features, (0, pad)), torch.nn.functional.pad(weight, (0, 0, 0, pad))
    return features, weight


def indice_conv(features, weight, bias, rulebook, inverse=False, keep_fp32=False):
    """keep_fp32: a layer that computes in fp32 under bf16 features (below) hands its fp32 result on as it is -- for a consumer that
    wants fp32 anyway (the occupancy head's dense maps): no rounding launch here, no widening launch there"""
    features, weight = _pad_in_channels(features, weight)
    if features.dtype == torch.bfloat16 and (weight.shape[-2] % 16 or weight.shape[-1] % 16):
        # bf16 activations exist in the LDS-DMA kernel only (channel counts that are multiples of 16); the few other layers
        # (4 / 6 / 34 input channels, 2 / 3-channel heads) run in fp32 and round their result
        out = indice_conv(features.float(), weight, bias, rulebook, inverse)
        return out if keep_fp32 else out.to(torch.bfloat16)
    if inverse:   # (inverse convs run on strided rulebooks: both maps exist)
        return SparseConvFunction.apply(features, weight, bias, rulebook.nbr_in, rulebook.nbr_out, rulebook.order_in, rulebook.order_out)
    return SparseConvFunction.apply(features, weight, bias, rulebook.nbr_out, rulebook.map_bwd, rulebook.order_out, rulebook.order_in)


def indice_conv_bn_relu(features, weight, bias, rulebook, bn, relu, inverse=False):
    """indice_conv followed by bn (a fusable BatchNorm1d, see fused_bn.fusable) and optionally ReLU, as one autograd node"""
    features, weight = _pad_in_channels(features, weight)
    if features.dtype == torch.bfloat16 and (weight.shape[-2] % 16 or weight.shape[-1] % 16):
        from . import fused_bn
        return fused_bn.batch_norm_relu(bn, indice_conv(features, weight, bias, rulebook, inverse), relu)
    training = bn.training or not bn.track_running_stats
    nbt = bn.num_batches_tracked if (bn.training and bn.track_running_stats) else None
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    maps = (rulebook.nbr_in, rulebook.nbr_out) if inverse else (rulebook.nbr_out, rulebook.map_bwd)
    ords = (rulebook.order_in, rulebook.order_out) if inverse else (rulebook.order_out, rulebook.order_in)
    F = fast() if (PROFILE is None and CAPTURE is None and NATIVE_AUTOGRAD) else None
    if F is not None and features.is_cuda:
        # C++ autograd node (csrc/binding.cpp ConvBNReLUNode): same launches, no Python Function.apply / ctx bookkeeping
        from . import fused_bn
        ws, need = fused_bn._ws(features.device, weight.shape[-1])
        return F.conv_bn_relu(_actc(features), _f32c(weight), bias, maps[0], maps[1], ords[0], ords[1], bn.weight, bn.bias, rm, rv, nbt, bool(training or rm is None),
                              float(bn.momentum), float(bn.eps), bool(relu), ws, need, bool(_overlap_ok(maps[0].shape[0])), bool(weight.is_leaf))
    return SparseConvBNReLUFunction.apply(features, weight, bias, maps[0], maps[1], bn.weight, bn.bias, rm, rv, nbt, training, bn.momentum,
                                          bn.eps, relu, ords[0], ords[1])


def indice_maxpool(features, rulebook):
    if features.dtype == torch.bfloat16:  # max of bf16 values is exact in either type
        return indice_maxpool(features.float(), rulebook).to(torch.bfloat16)
    return SparseMaxPoolFunction.apply(features, rulebook.nbr_out, rulebook.nbr_in)
