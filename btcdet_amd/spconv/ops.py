"""Operator layer: rulebook construction and the autograd functions over libbtcdet_hip.so.

Mirrors ``spconv.ops`` / ``spconv.functional`` of spconv v1.2.1: the
reference reaches these through SubMConv3d / SparseConv3d / SparseConvTranspose3d /
SparseInverseConv3d / SparseMaxPool3d (/root/reference/btcdet/models/backbones_3d/
spconv_backbone.py:12-29).  The rulebook is kept as two dense neighbour maps (see
include/btcdet_hip.h); ``Rulebook.indice_pairs()`` gives spconv's (2,K,N)/(K,) view on demand.
"""
import numpy as np
import torch

from .. import _lib
from .._lib import MODE_CONV, MODE_SUBM, MODE_TRANSPOSE, check, i3, i3p, lib, ptr, stream_ptr, workspace


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


def _span(name, nbytes_fn):
    if PROFILE is None:
        return _NOSPAN
    r = nbytes_fn()
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
    """One cached rulebook (what spconv stores in ``indice_dict[indice_key]``)."""

    __slots__ = ("out_indices", "in_indices", "nbr_out", "nbr_in", "in_shape", "out_shape", "K", "mode")

    def __init__(self, out_indices, in_indices, nbr_out, nbr_in, in_shape, out_shape, K, mode):
        self.out_indices, self.in_indices = out_indices, in_indices
        self.nbr_out, self.nbr_in = nbr_out, nbr_in
        self.in_shape, self.out_shape = list(in_shape), list(out_shape)
        self.K, self.mode = K, mode

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
        check(lib().btc_pairs_from_nbr(ptr(self.nbr_out), self.n_out, self.K, self.n_in, ptr(pairs), ptr(num),
                                       stream_ptr()), "btc_pairs_from_nbr")
        return pairs[:, :, :self.n_in], num


def _as_idx(indices):
    if indices.dtype != torch.int32:
        indices = indices.int()
    return indices.contiguous()


def build_rulebook(indices, batch_size, spatial_shape, ksize, stride=1, padding=0, dilation=1, out_padding=0,
                   subm=False, transpose=False):
    """ops.get_indice_pairs replacement.  indices (N,4) int32 [b,z,y,x] on the GPU."""
    indices = _as_idx(indices)
    if indices.dim() != 2 or indices.shape[1] != 4:
        raise _lib.BtcHipError(f"indices must be (N,4) [b,z,y,x], got {tuple(indices.shape)}")
    dev = indices.device
    n = indices.shape[0]
    L = lib()
    k3, s3, p3, d3, op3 = i3(ksize), i3(stride), i3(padding), i3(dilation), i3(out_padding)
    in_sh = i3([int(v) for v in spatial_shape])
    K = int(np.prod(k3))
    mode = MODE_SUBM if subm else (MODE_TRANSPOSE if transpose else MODE_CONV)
    out_sh = np.zeros(3, dtype=np.int32)
    check(L.btc_out_shape(i3p(in_sh), i3p(k3), i3p(s3), i3p(p3), i3p(d3), i3p(op3), mode, i3p(out_sh)), "btc_out_shape")
    if PROFILE is not None:
        return _build_rulebook_profiled(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, out_padding,
                                        subm, transpose)
    return _build_rulebook(indices, batch_size, in_sh, out_sh, k3, s3, p3, d3, K, mode, subm)


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


def _build_rulebook(indices, batch_size, in_sh, out_sh, k3, s3, p3, d3, K, mode, subm):
    dev = indices.device
    n = indices.shape[0]
    L = lib()
    if subm:
        nbr_out = torch.empty((n, K), dtype=torch.int32, device=dev)
        nbr_in = torch.empty((n, K), dtype=torch.int32, device=dev)
        ws_bytes = L.btc_rulebook_subm_ws_bytes(n)
        ws = workspace(ws_bytes, dev)
        check(L.btc_rulebook_subm(ptr(indices), n, int(batch_size), i3p(in_sh), i3p(k3), i3p(d3), ptr(nbr_out),
                                  ptr(nbr_in), ptr(ws), ws_bytes, stream_ptr()), "btc_rulebook_subm")
        return Rulebook(indices, indices, nbr_out, nbr_in, in_sh.tolist(), out_sh.tolist(), K, mode)
    ws_bytes = L.btc_rulebook_conv_ws_bytes(int(batch_size), i3p(out_sh))
    ws = workspace(ws_bytes, dev)
    d_n_out = torch.empty((1,), dtype=torch.int32, device=dev)
    check(L.btc_rulebook_conv_count(ptr(indices), n, int(batch_size), i3p(in_sh), i3p(out_sh), i3p(k3), i3p(s3),
                                    i3p(p3), i3p(d3), mode, ptr(d_n_out), ptr(ws), ws_bytes, stream_ptr()),
          "btc_rulebook_conv_count")
    n_out = int(d_n_out.item())  # the one read-back of the build (spconv syncs here too)
    out_indices = torch.empty((n_out, 4), dtype=torch.int32, device=dev)
    nbr_out = torch.empty((n_out, K), dtype=torch.int32, device=dev)
    nbr_in = torch.empty((n, K), dtype=torch.int32, device=dev)
    check(L.btc_rulebook_conv_fill(ptr(indices), n, int(batch_size), i3p(in_sh), i3p(out_sh), i3p(k3), i3p(s3),
                                   i3p(p3), i3p(d3), mode, n_out, ptr(out_indices), ptr(nbr_out), ptr(nbr_in), ptr(ws),
                                   ws_bytes, stream_ptr()), "btc_rulebook_conv_fill")
    return Rulebook(out_indices, indices, nbr_out, nbr_in, in_sh.tolist(), out_sh.tolist(), K, mode)


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


class SparseConvFunction(torch.autograd.Function):
    """indice_conv / indice_subm_conv / indice_inverse_conv in one function.
    map_fwd (n_res,K): source row gathered by result row i at offset k; map_bwd (n_src,K) its transpose."""

    @staticmethod
    def forward(ctx, features, weight, bias, map_fwd, map_bwd):
        features = _actc(features)
        bf = features.dtype == torch.bfloat16
        w = _f32c(weight)
        cin, cout = w.shape[-2], w.shape[-1]
        K = map_fwd.shape[1]
        if w.numel() != K * cin * cout or features.shape[1] != cin:
            raise _lib.BtcHipError(f"weight {tuple(w.shape)} does not match K={K}, Cin={features.shape[1]}")
        n_res = map_fwd.shape[0]
        out = torch.empty((n_res, cout), dtype=features.dtype, device=features.device)
        b = _f32c(bias) if bias is not None else None
        fwd = lib().btc_conv_fwd_bf16 if bf else lib().btc_conv_fwd
        with _span("conv_apply", lambda: _conv_cost(map_fwd, n_res, K, cin, cout, 2 if bf else 4)):
            check(fwd(ptr(features), ptr(w), ptr(b), ptr(map_fwd), n_res, K, cin, cout, ptr(out), stream_ptr()), "btc_conv_fwd")
        if CAPTURE is not None:
            CAPTURE.append((features, w, b, map_fwd, map_bwd))
        ctx.save_for_backward(features, w, map_fwd, map_bwd)
        ctx.has_bias = bias is not None
        ctx.wshape = tuple(weight.shape)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        features, w, map_fwd, map_bwd = ctx.saved_tensors
        bf = features.dtype == torch.bfloat16
        grad_out = _actc(grad_out if grad_out.dtype == features.dtype else grad_out.to(features.dtype))
        cin, cout = w.shape[-2], w.shape[-1]
        K = map_fwd.shape[1]
        L = lib()
        wgrad, dgrad = (L.btc_conv_wgrad_bf16, L.btc_conv_dgrad_bf16) if bf else (L.btc_conv_wgrad, L.btc_conv_dgrad)
        din = dw = db = None
        need_din, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        # dgrad and wgrad are independent and each is latency bound on its own at BtcDet's sizes: wgrad goes to a side
        # HIP stream (fork / join with events, no host sync) so the two kernels share the GPU
        side = _side_stream(grad_out.device) if (need_din and need_dw and OVERLAP_WGRAD and PROFILE is None) else None
        if need_dw:
            n_res, n_src = map_fwd.shape[0], map_bwd.shape[0]
            ws_bytes = L.btc_conv_wgrad_ws_bytes(n_res, K, cin, cout, n_src)
            if side is not None:
                main = torch.cuda.current_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    dw = torch.empty(ctx.wshape, dtype=torch.float32, device=grad_out.device)
                    ws = workspace(ws_bytes, grad_out.device)
                    check(wgrad(ptr(features), ptr(grad_out), ptr(map_fwd), n_res, ptr(map_bwd), n_src, K, cin, cout,
                                           ptr(dw), ptr(ws), ws_bytes, stream_ptr()), "btc_conv_wgrad")
                for t in (features, grad_out, map_fwd, map_bwd):
                    t.record_stream(side)
            else:
                dw = torch.empty(ctx.wshape, dtype=torch.float32, device=grad_out.device)
                ws = workspace(ws_bytes, grad_out.device)
                with _span("conv_wgrad", lambda: _wgrad_cost(map_fwd, n_res, K, cin, cout, 2 if bf else 4)):
                    check(wgrad(ptr(features), ptr(grad_out), ptr(map_fwd), n_res, ptr(map_bwd), n_src, K, cin, cout,
                                           ptr(dw), ptr(ws), ws_bytes, stream_ptr()), "btc_conv_wgrad")
        if need_din:
            n_src = map_bwd.shape[0]
            din = torch.empty((n_src, cin), dtype=features.dtype, device=grad_out.device)
            with _span("conv_apply", lambda: _conv_cost(map_bwd, n_src, K, cout, cin, 2 if bf else 4)):
                check(dgrad(ptr(grad_out), ptr(w), ptr(map_bwd), n_src, K, cin, cout, ptr(din), stream_ptr()),
                      "btc_conv_dgrad")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = grad_out.sum(0, dtype=torch.float32)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)  # join: dW is consumed on the main stream from here on
            dw.record_stream(torch.cuda.current_stream())
        return din, dw, db, None, None


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


def indice_conv(features, weight, bias, rulebook, inverse=False):
    if features.dtype == torch.bfloat16 and (weight.shape[-2] % 16 or weight.shape[-1] % 16):
        # bf16 activations exist in the LDS-DMA kernel only (channel counts that are multiples of 16); the few other layers
        # (4 / 6 / 34 input channels, 2 / 3-channel heads) run in fp32 and round their result
        return indice_conv(features.float(), weight, bias, rulebook, inverse).to(torch.bfloat16)
    if inverse:
        return SparseConvFunction.apply(features, weight, bias, rulebook.nbr_in, rulebook.nbr_out)
    return SparseConvFunction.apply(features, weight, bias, rulebook.nbr_out, rulebook.nbr_in)


def indice_maxpool(features, rulebook):
    if features.dtype == torch.bfloat16:  # max of bf16 values is exact in either type
        return indice_maxpool(features.float(), rulebook).to(torch.bfloat16)
    return SparseMaxPoolFunction.apply(features, rulebook.nbr_out, rulebook.nbr_in)
