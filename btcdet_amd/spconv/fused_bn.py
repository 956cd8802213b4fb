"""Fused BatchNorm1d(+ReLU) over `.features` -- used by SparseSequential when it meets the reference's
``norm_fn(out_channels), nn.ReLU()`` pair (spconv_backbone.py:33-43).  The modules, their parameters, buffers and
state_dict keys stay the plain torch ones; only the arithmetic goes through btc_bn_relu_fwd / btc_bn_relu_bwd."""
import torch

from .._lib import check, fast, lib, ptr, stream_ptr

_WS = {}


def _ws(device, C):
    """persistent workspace per (device, current stream): its head (arrival counter) starts zeroed and every call leaves it
    zeroed; calls on one stream are serialised, calls on different streams (the occupancy branch's backward beside the
    detection branch's forward, bench.make_step) must not share the partial sums"""
    need = lib().btc_bn_ws_bytes(int(C))
    key = (device.index, stream_ptr()) if device.type == "cuda" else (device.type, device.index)
    buf = _WS.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.zeros(int(lib().btc_bn_ws_bytes(max(int(C), 1024))), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf, need


_FUSE_WS = {}


def fuse_ws(device):
    """zero-initialised slot buffer of the fused conv + BatchNorm statistics (csrc/bn_fuse.h), one per (device, current stream): the
    kernels leave it zeroed; launches on one stream are serialised, streams must not share it"""
    key = (device.index, stream_ptr())
    buf = _FUSE_WS.get(key)
    if buf is None:
        buf = _FUSE_WS[key] = torch.zeros(int(lib().btc_bn_fuse_ws_bytes()), dtype=torch.uint8, device=device)
    return buf


def conv_bn_forward(features, w, b, map_fwd, ord_fwd, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, relu):
    """training-mode conv -> BatchNorm -> [ReLU] through btc_conv_bn_relu_fwd (statistics in the conv's epilogue); fp32 weights
    or their split planes.
    -> (x, y, stats (2, C) = mean | rstd)"""
    n, K = map_fwd.shape
    cin, cout = w.shape[-2], w.shape[-1]
    x = torch.empty((n, cout), dtype=features.dtype, device=features.device)
    y = torch.empty_like(x)
    stats = torch.empty((2, cout), dtype=torch.float32, device=features.device)
    ws, need = _ws(features.device, cout)
    operands = 1 if features.dtype == torch.bfloat16 else 0
    from . import ops
    if ops._split_operands(features, K, cin, cout, n):   # split-operand kernel: W = the forward planes
        operands, w = 3, ops._weights_split(w, K, cin, cout)[1]
    check(lib().btc_conv_bn_relu_fwd(operands, ptr(features), ptr(w), ptr(b), ptr(map_fwd), ptr(ord_fwd), n, K, cin, cout,
                                     ptr(x), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ptr(num_batches_tracked), float(momentum),
                                     float(eps), int(relu), ptr(y), ptr(stats[0]), ptr(stats[1]), ptr(ws), need, ptr(fuse_ws(features.device)),
                                     stream_ptr()), "btc_conv_bn_relu_fwd")
    return x, y, stats


def bn_forward(x, weight, bias, running_mean, running_var, num_batches_tracked, use_batch, momentum, eps, relu):
    """y = [relu](batchnorm(x)); returns (y, stats) with stats (2, C) = mean | rstd.  Running statistics / num_batches_tracked
    are updated in the kernel when given (training)."""
    N, C = x.shape
    ws, need = _ws(x.device, C)
    F = fast()
    if F is not None:
        return F.bn_fwd(x, weight, bias, running_mean, running_var, num_batches_tracked, bool(use_batch), float(momentum), float(eps),
                        bool(relu), ws, need, stream_ptr())
    y = torch.empty_like(x)
    stats = torch.empty((2, C), dtype=torch.float32, device=x.device)
    fwd = lib().btc_bn_relu_fwd_bf16 if x.dtype == torch.bfloat16 else lib().btc_bn_relu_fwd
    check(fwd(ptr(x), N, C, ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var), ptr(num_batches_tracked), float(momentum),
              float(eps), int(use_batch), int(relu), ptr(y), ptr(stats[0]), ptr(stats[1]), ptr(ws), need, stream_ptr()), "btc_bn_relu_fwd")
    return y, stats


def bn_backward(x, y, dy, weight, stats, use_batch, relu):
    """-> dx, dgamma, dbeta (the last two are rows of one (2, C) tensor)"""
    N, C = x.shape
    ws, need = _ws(x.device, C)
    F = fast()
    if F is not None:
        dx, dparam = F.bn_bwd(x, y, dy, weight, stats, bool(use_batch), bool(relu), ws, need, stream_ptr())
        return dx, dparam[0], dparam[1]
    dx = torch.empty_like(x)
    dparam = torch.empty((2, C), dtype=torch.float32, device=x.device)
    bwd = lib().btc_bn_relu_bwd_bf16 if x.dtype == torch.bfloat16 else lib().btc_bn_relu_bwd
    check(bwd(ptr(x), ptr(y), ptr(dy), N, C, ptr(weight), ptr(stats[0]), ptr(stats[1]), int(use_batch), int(relu), ptr(dx), ptr(dparam[0]),
              ptr(dparam[1]), ptr(ws), need, stream_ptr()), "btc_bn_relu_bwd")
    return dx, dparam[0], dparam[1]


def col_sum(x):
    """x.sum(0) of a contiguous (N, C) fp32 / bf16 CUDA matrix in fp32 (one launch, fp64 accumulation in a fixed order):
    the bias gradient of a sparse conv.  torch's column reduction of a 200 K x 5 matrix runs on 2 workgroups (120 us)."""
    N, C = x.shape
    ws, need = _ws(x.device, C)
    out = torch.empty((C,), dtype=torch.float32, device=x.device)
    fn = lib().btc_col_sum_bf16 if x.dtype == torch.bfloat16 else lib().btc_col_sum
    check(fn(ptr(x), N, C, ptr(out), ptr(ws), need, stream_ptr()), "btc_col_sum")
    return out


class BatchNormReLUFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, num_batches_tracked, training, momentum, eps, relu):
        x = x.contiguous()
        use_batch = bool(training or running_mean is None)
        y, stats = bn_forward(x, weight, bias, running_mean, running_var, num_batches_tracked if training else None, use_batch,
                              momentum, eps, relu)
        ctx.save_for_backward(x, y, weight, stats)
        ctx.flags = (use_batch, bool(relu))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, stats = ctx.saved_tensors
        use_batch, relu = ctx.flags
        dy = (dy if dy.dtype == x.dtype else dy.to(x.dtype)).contiguous()
        dx, dgamma, dbeta = bn_backward(x, y, dy, weight, stats, use_batch, relu)
        return dx, (dgamma if weight is not None else None), (dbeta if weight is not None else None), None, None, None, None, None, None, None


def fusable(bn):
    return isinstance(bn, torch.nn.BatchNorm1d) and bn.affine == (bn.weight is not None) and \
        (bn.momentum is not None) and (bn.track_running_stats or bn.training)


def batch_norm_relu(bn, x, relu):
    """same semantics as bn(x) followed by ReLU for a (N,C) float32 GPU tensor with N >= 1"""
    training = bn.training or not bn.track_running_stats
    nbt = bn.num_batches_tracked if (bn.training and bn.track_running_stats) else None
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return BatchNormReLUFunction.apply(x, bn.weight, bn.bias, rm, rv, nbt, training, bn.momentum, bn.eps, relu)
