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
    """training-mode conv -> BatchNorm -> [ReLU] through btc_conv_bn_relu_fwd (statistics in the conv's epilogue); fp32 weights,
    their split planes, or (bf16 features) their bf16 copy.
    -> (x, y, stats (2, C) = mean | rstd)"""
    n, K = map_fwd.shape
    cin, cout = w.shape[-2], w.shape[-1]
    x = torch.empty((n, cout), dtype=features.dtype, device=features.device)
    y = torch.empty_like(x)
    stats = torch.empty((2, cout), dtype=torch.float32, device=features.device)
    ws, need = _ws(features.device, cout)
    operands = 1 if features.dtype == torch.bfloat16 else 0
    from . import ops
    if ops._bf16_operands(features, K, cin, cout):       # bf16-operand kernel: W = the forward (transposed) bf16 copy
        operands, w = 2, ops._weights_bf16(w, K, cin, cout)[1]
    elif ops._split_operands(features, K, cin, cout, n):   # split-operand kernel: W = the forward planes
        operands, w = 3, ops._weights_split(w, K, cin, cout)[1]
    check(lib().btc_conv_bn_relu_fwd_src(operands, ptr(features), int(features.shape[0]), ptr(w), ptr(b), ptr(map_fwd), ptr(ord_fwd), n, K, cin, cout,
                                     ptr(x), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ptr(num_batches_tracked), float(momentum),
                                     float(eps), int(relu), ptr(y), ptr(stats[0]), ptr(stats[1]), ptr(ws), need, ptr(fuse_ws(features.device)),
                                     stream_ptr()), "btc_conv_bn_relu_fwd_src")
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
    """this BatchNorm1d, in its current state, through the rank-local fused kernels?  (Not when it synchronises its batch statistics
    over a process group: is_sync)"""
    return isinstance(bn, torch.nn.BatchNorm1d) and bn.affine == (bn.weight is not None) and \
        (bn.momentum is not None) and (bn.track_running_stats or bn.training) and not is_sync(bn)


# ----------------------------------------------------------------------------------------------------------------------
# --sync_bn (/root/reference/tools/train.py:32,130-131: torch.nn.SyncBatchNorm.convert_sync_batchnorm): the batch statistics of a
# BatchNorm1d over sparse features taken over ALL ranks' rows.  convert_sync_batchnorm() below marks the modules (same objects, same
# parameters, buffers and state_dict keys); SparseSequential then routes them here.  Per layer and step, as torch's SyncBatchNorm:
# one all_gather of (mean, biased var, count) forward, one all_reduce of (sum dy x^, sum dy) backward -- the normalisation and its
# backward are the fused HIP kernels in their given-statistics mode (btc_bn_relu_fwd / _bwd, training = 0) plus one elementwise
# correction for the two mean terms.
# ----------------------------------------------------------------------------------------------------------------------
def is_sync(bn):
    return getattr(bn, "sync_group", None) is not None and bn.training


def _routed_here(parent, child):
    """is `child` a BatchNorm1d whose forward goes through this module's kernels -- a direct member of a SparseSequential, or bn1 / bn2
    of a residual block (backbones_3d._bn_act)?  Only those can be synchronised by marking."""
    from .modules import SparseSequential
    if not (isinstance(child, torch.nn.BatchNorm1d) and child.affine and child.momentum is not None):
        return False
    return isinstance(parent, SparseSequential) or type(parent).__name__ == "SparseBasicBlock"


def convert_sync_batchnorm(module, process_group=None):
    """torch.nn.SyncBatchNorm.convert_sync_batchnorm for a model of this package (/root/reference/tools/train.py:130-131 converts EVERY
    _BatchNorm): BatchNorm1d layers over sparse features (members of a SparseSequential / a residual block) are MARKED -- same objects,
    parameters, buffers, state_dict keys -- and take their batch statistics over process_group (default: the world) while in training
    mode; every other _BatchNorm under `module` (the BatchNorm2d layers of BaseBEVBackbone, ConvHead's and pointnet2_stack's 1-D / 2-D
    ones) is replaced in its parent by torch's own SyncBatchNorm (which adopts the module's parameters and buffers: same keys).  A world
    of one rank leaves the module alone.  -> number of modules that now synchronise (marked + replaced)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return 0
    group = process_group if process_group is not None else dist.group.WORLD
    n = 0
    if isinstance(module, torch.nn.BatchNorm1d) and module.affine and module.momentum is not None:
        module.sync_group = group      # a bare layer: its caller routes it (sync_batch_norm_relu); it cannot be replaced in a parent
        return 1
    for parent in list(module.modules()):
        for name, child in list(parent.named_children()):
            if isinstance(child, torch.nn.SyncBatchNorm):
                continue
            if _routed_here(parent, child):
                child.sync_group = group
                n += 1
            elif isinstance(child, torch.nn.modules.batchnorm._BatchNorm):
                setattr(parent, name, torch.nn.SyncBatchNorm.convert_sync_batchnorm(child, process_group))
                n += 1
    return n


def combine_stats(gathered, C):
    """gathered (W, 2C + 1): per rank (mean | biased var | row count) -> (mean, biased var, total count) of the concatenated batch
    (Chan et al.'s pairwise update, all ranks at once; ranks without rows carry count 0; no rows on ANY rank: zeros, count 0)"""
    cnt = gathered[:, 2 * C:2 * C + 1]
    n = cnt.sum()
    d = n.clamp(min=1.0)
    mean = (gathered[:, :C] * cnt).sum(0) / d
    var = ((gathered[:, C:2 * C] + (gathered[:, :C] - mean) ** 2) * cnt).sum(0) / d
    return mean, var, n


def _via_host(group):
    import torch.distributed as dist
    return dist.get_backend(group) == "gloo"   # (2 ranks on ONE GPU in the tests: device tensors staged through the host)


def _all_gather_rows(t, group):
    import torch.distributed as dist
    W = dist.get_world_size(group)
    src = t.cpu() if _via_host(group) else t
    out = torch.empty((W,) + tuple(src.shape), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src.contiguous(), group=group) if not _via_host(group) else \
        dist.all_gather(list(out.unbind(0)), src.contiguous(), group=group)
    return out.to(t.device)


def _all_reduce_sum(t, group):
    import torch.distributed as dist
    if _via_host(group):
        h = t.cpu()
        dist.all_reduce(h, group=group)
        return h.to(t.device)
    dist.all_reduce(t, group=group)
    return t


class SyncBatchNormReLUFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, relu, group):
        x = x.contiguous()
        N, C = x.shape
        xf = x if x.dtype == torch.float32 else x.float()
        if N > 0:
            var_l, mean_l = torch.var_mean(xf, dim=0, unbiased=False)
        else:
            var_l = mean_l = torch.zeros((C,), dtype=torch.float32, device=x.device)
        mine = torch.cat([mean_l, var_l, torch.full((1,), float(N), dtype=torch.float32, device=x.device)])
        mean, var, n = combine_stats(_all_gather_rows(mine, group), C)
        if running_mean is not None:   # torch's update: unbiased variance of the WHOLE batch (a batch without rows on any rank moves nothing)
            with torch.no_grad():
                m = momentum * (n > 0).to(mean.dtype)
                running_mean.mul_(1.0 - m).add_(mean * m)
                running_var.mul_(1.0 - m).add_(var * (n / (n - 1.0).clamp_(min=1.0)) * m)
                if num_batches_tracked is not None:
                    num_batches_tracked.add_(1)
        # normalisation (+ ReLU) with GIVEN statistics: the fused kernel's eval path
        y, stats = bn_forward(x, weight, bias, mean.contiguous(), var.contiguous(), None, False, momentum, eps, relu) if N > 0 else \
            (torch.empty_like(x), torch.stack([mean, torch.rsqrt(var + eps)]))
        ctx.save_for_backward(x, y, weight, stats, n)
        ctx.meta = (bool(relu), group)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, stats, n = ctx.saved_tensors
        relu, group = ctx.meta
        N, C = x.shape
        if N > 0:
            dy = (dy if dy.dtype == x.dtype else dy.to(x.dtype)).contiguous()
            dx_e, dgamma, dbeta = bn_backward(x, y, dy, weight, stats, False, relu)     # dx_e = gamma rstd dy', dgamma = sum dy' x^, dbeta = sum dy'
        else:
            dx_e, dgamma, dbeta = torch.empty_like(x), torch.zeros_like(weight), torch.zeros_like(weight)
        tot = _all_reduce_sum(torch.stack([dgamma, dbeta]), group)
        if N > 0:
            mean, rstd = stats[0], stats[1]
            a = weight * rstd * rstd * tot[0] / n
            b = weight * rstd * tot[1] / n - mean * a
            dx = torch.addcmul(dx_e.float() - b, x.float(), a, value=-1.0).to(x.dtype)   # - gamma rstd (mean(dy') + x^ mean(dy' x^)) over ALL ranks' rows
        else:
            dx = dx_e
        return dx, dgamma, dbeta, None, None, None, None, None, None, None


def sync_batch_norm_relu(bn, x, relu):
    """training-mode bn(x) [+ ReLU] with the statistics of all ranks' rows (is_sync(bn)); (N, C) float32 / bfloat16 GPU tensor"""
    tr = bn.track_running_stats
    return SyncBatchNormReLUFunction.apply(x, bn.weight, bn.bias, bn.running_mean if tr else None, bn.running_var if tr else None,
                                           bn.num_batches_tracked if tr else None, bn.momentum, bn.eps, relu, bn.sync_group)


def batch_norm_relu(bn, x, relu):
    """same semantics as bn(x) followed by ReLU for a (N,C) float32 GPU tensor with N >= 1"""
    training = bn.training or not bn.track_running_stats
    nbt = bn.num_batches_tracked if (bn.training and bn.track_running_stats) else None
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return BatchNormReLUFunction.apply(x, bn.weight, bn.bias, rm, rv, nbt, training, bn.momentum, bn.eps, relu)
