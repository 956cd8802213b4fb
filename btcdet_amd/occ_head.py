"""OccHead3D: occupancy probability / residual head of the occupancy branch and its losses.

Mirrors /root/reference/btcdet/models/occ_pnt/occ_dense_heads/occ_head_3D.py:10-52 and
occ_head_template.py:8-111 (masked focal / smooth-L1 means) with the loss functions of
/root/reference/btcdet/utils/loss_utils.py:77-169 (softmax focal, eps 1e-6 as passed at :169) and
:174-240 (smooth L1, beta = res_beta)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import spconv


class SoftmaxFocalClassificationLoss(nn.Module):
    def __init__(self, alpha=1.0, gamma=2.0, reduction='none'):
        super().__init__()
        self.alpha, self.gamma, self.reduction, self.eps = alpha, gamma, reduction, 1e-6

    def forward(self, input, target, weights):
        p = F.softmax(input, dim=1) + self.eps
        focal = -self.alpha * torch.pow(-p + 1., self.gamma) * torch.log(p)
        loss = torch.sum(target * focal, dim=1, keepdim=True)
        return loss if weights is None else loss * weights


class SigmoidFocalClassificationLoss(nn.Module):
    def __init__(self, gamma=2.0, alpha=0.25):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma

    def forward(self, input, target, weights):
        p = torch.sigmoid(input)
        alpha_w = target * self.alpha + (1 - target) * (1 - self.alpha)
        pt = target * (1.0 - p) + (1.0 - target) * p
        bce = torch.clamp(input, min=0) - input * target + torch.log1p(torch.exp(-torch.abs(input)))
        loss = alpha_w * torch.pow(pt, self.gamma) * bce
        if weights is None:
            return loss
        if weights.dim() == 2 or (weights.dim() == 1 and target.dim() == 2):
            weights = weights.unsqueeze(-1)
        return loss * weights


class WeightedSmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0 / 9.0, code_weights=None):
        super().__init__()
        self.beta = beta

    def forward(self, input, target, weights=None):
        target = torch.where(torch.isnan(target), input, target)
        n = torch.abs(input - target)
        loss = n if self.beta < 1e-5 else torch.where(n < self.beta, 0.5 * n ** 2 / self.beta, n - 0.5 * self.beta)
        if weights is not None:
            loss = loss * weights.unsqueeze(-1)
        return loss


_LOSS_WS = {}


class OccLossFunction(torch.autograd.Function):
    """(cls_loss, reg_loss) of the occupancy head in one launch each way (csrc/occ_loss.hip)"""

    @staticmethod
    def forward(ctx, logit, res, res_target, pos_mask, cls_mask, cls_w, reg_mask, reg_w, beta, w_cls, w_res):
        from ._lib import check, lib, ptr, stream_ptr
        dev = logit.device
        key = (dev.type, dev.index)
        if key not in _LOSS_WS:
            _LOSS_WS[key] = torch.zeros(int(lib().btc_occ_loss_ws_bytes()), dtype=torch.uint8, device=dev)
        ws = _LOSS_WS[key]
        B = logit.shape[0]
        ncell = logit[0, 0].numel()
        logit = logit.contiguous()
        res_c = res.contiguous() if res is not None else None
        tgt = res_target.contiguous() if res_target is not None else None
        u8 = lambda t: (t if t.dtype == torch.uint8 else t.to(torch.uint8)).contiguous()
        pos_mask, cls_mask = u8(pos_mask), u8(cls_mask)
        reg_mask = u8(reg_mask) if reg_mask is not None else None
        cls_w = cls_w.contiguous()
        reg_w = reg_w.contiguous() if reg_w is not None else None
        out = torch.empty((3,), dtype=torch.float32, device=dev)       # cls | reg | cls + reg
        norms = torch.empty((2,), dtype=torch.float32, device=dev)
        check(lib().btc_occ_loss_fwd_total(ptr(logit), ptr(res_c), ptr(tgt), ptr(pos_mask), ptr(cls_mask), ptr(cls_w), ptr(reg_mask), ptr(reg_w), B,
                                           ncell, float(beta), float(w_cls), float(w_res), ptr(out), ptr(norms), ptr(ws), ws.numel(), stream_ptr()),
              "btc_occ_loss_fwd_total")
        ctx.save_for_backward(logit, res_c, tgt, pos_mask, cls_mask, cls_w, reg_mask, reg_w, norms)
        ctx.meta = (B, ncell, float(beta))
        # -> (the loss the head returns: cls + reg, made by the kernel; the two parts for logging).  The reference adds them with one more
        # launch and autograd splits the gradient back through two select_backward fills + copies and an add: 7 launches of nothing.
        parts = out[:2]
        ctx.mark_non_differentiable(parts)
        return out[2], parts

    @staticmethod
    def backward(ctx, grad_total, _grad_parts):
        from ._lib import check, lib, ptr, stream_ptr
        logit, res_c, tgt, pos_mask, cls_mask, cls_w, reg_mask, reg_w, norms = ctx.saved_tensors
        B, ncell, beta = ctx.meta
        d_logit = torch.empty_like(logit)                                  # (the kernel writes every cell: no fills)
        d_res = torch.empty_like(res_c) if res_c is not None else None
        g = grad_total.reshape(1).to(torch.float32).contiguous()
        check(lib().btc_occ_loss_bwd_total(ptr(logit), ptr(res_c), ptr(tgt), ptr(pos_mask), ptr(cls_mask), ptr(cls_w), ptr(reg_mask), ptr(reg_w), B,
                                           ncell, beta, ptr(norms), ptr(g), ptr(d_logit), ptr(d_res), stream_ptr()), "btc_occ_loss_bwd_total")
        return d_logit, d_res, None, None, None, None, None, None, None, None, None


class LazyScalar(object):
    """a float that is still on its way from the GPU: reads (float(), '%f', format, comparisons, arithmetic) wait for the copy"""
    __slots__ = ("_host", "_i", "_event", "_value")

    def __init__(self, host, i, event):
        self._host, self._i, self._event, self._value = host, i, event, None

    def item(self):
        if self._value is None:
            self._event.synchronize()
            self._value = float(self._host[self._i])
            self._host = self._event = None
        return self._value

    __float__ = item

    def __repr__(self):
        return repr(self.item())

    def __format__(self, spec):
        return format(self.item(), spec)

    def __array__(self, dtype=None, copy=None):  # np.asarray([...LazyScalar...]) -> float array, not an object array
        import numpy as np
        return np.asarray(self.item(), dtype=dtype)

    def __eq__(self, o): return self.item() == float(o)
    def __lt__(self, o): return self.item() < float(o)
    def __le__(self, o): return self.item() <= float(o)
    def __gt__(self, o): return self.item() > float(o)
    def __ge__(self, o): return self.item() >= float(o)
    def __hash__(self): return hash(self.item())
    def __add__(self, o): return self.item() + float(o)
    __radd__ = __add__
    def __sub__(self, o): return self.item() - float(o)
    def __rsub__(self, o): return float(o) - self.item()
    def __mul__(self, o): return self.item() * float(o)
    __rmul__ = __mul__
    def __truediv__(self, o): return self.item() / float(o)
    def __rtruediv__(self, o): return float(o) / self.item()
    def __neg__(self): return -self.item()
    def __abs__(self): return abs(self.item())
    def __bool__(self): return bool(self.item())


def _lazy_scalars(t, n):
    """first n elements of a small fp32 tensor as host numbers; CUDA tensors: asynchronous copy + LazyScalar"""
    if not t.is_cuda:
        return t.tolist()[:n]
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return [LazyScalar(host, i, ev) for i in range(n)]


MERGE_HEADS = True  # OccHead3D: conv_cls + conv_res as one launch per direction
FUSED_LOSS = True  # OccHeadTemplate.get_loss through btc_occ_loss_* (False: the torch op chain of the reference)


def _join_wgrad_hook(grad):
    from .spconv import ops
    ops.join_wgrad()
    return None


class OccHeadTemplate(nn.Module):
    def __init__(self, model_cfg, data_cfg, num_class, grid_size):
        super().__init__()
        self.data_cfg, self.model_cfg = data_cfg, model_cfg
        head = self.model_cfg.OCC_DENSE_HEAD
        self.noloss = bool(head.get("NOLOSS", None))
        self.num_class = num_class
        self.forward_ret_dict = {}
        self.nx, self.ny, self.nz = grid_size
        lw = head.LOSS_CONFIG.LOSS_WEIGHTS
        self.occ_fore_res_weight = lw.get("occ_fore_res_weight", 0.1)
        self.occ_fore_cls_weight = lw.get("occ_fore_cls_weight", 1.0)
        self.res_num_dim = data_cfg.OCC.RES_NUM_DIM
        self.reg = model_cfg.PARAMS.get("REG", False)
        self.build_losses(head.LOSS_CONFIG)

    def prepare_loss_map(self, batch_dict):
        return batch_dict

    def build_losses(self, losses_cfg):
        if self.noloss:
            return
        if self.is_softmax:
            self.add_module('cls_loss_func', SoftmaxFocalClassificationLoss(alpha=1.0, gamma=2.0))
        else:
            self.add_module('cls_loss_func', SigmoidFocalClassificationLoss(alpha=losses_cfg.LOSS_WEIGHTS['cls_alpha'], gamma=2.0))
        if self.reg:
            self.add_module('reg_loss_func', WeightedSmoothL1Loss(beta=losses_cfg.LOSS_WEIGHTS['res_beta'],
                                                                  code_weights=[1.0] * self.res_num_dim))

    @staticmethod
    def mean_masked_loss(pred, target, loss_func, loss_weight_float, mask=None):
        """weighted mean of the per-cell loss over the cells of `mask` (occ_head_template.py:96-108)"""
        inds = mask.nonzero() if mask is not None else (loss_weight_float > 1e-4).nonzero()
        b, z, y, x = inds[:, 0], inds[:, 1], inds[:, 2], inds[:, 3]
        w = loss_weight_float[b, :, z, y, x]
        loss = loss_func(pred[b, :, z, y, x], target[b, :, z, y, x], weights=None) * w
        return torch.sum(loss) / torch.clamp(torch.sum(w), min=1.0)

    def get_cls_layer_loss(self, batch_dict):
        w = batch_dict["general_cls_loss_mask_float"].unsqueeze(1)
        logit = batch_dict['pred_occ_logit']
        pos = batch_dict["pos_mask"].to(logit.dtype)
        onehot = torch.stack([1.0 - pos, pos], dim=-1)
        onehot = (onehot if self.is_softmax else onehot[..., 1:]).permute(0, 4, 1, 2, 3)
        loss = self.mean_masked_loss(logit, onehot, self.cls_loss_func, w, mask=batch_dict['general_cls_loss_mask'])
        loss = loss * self.occ_fore_cls_weight
        return loss, {'occ_loss_cls': loss.item()}

    def get_res_layer_loss(self, batch_dict):
        w = batch_dict["general_reg_loss_mask_float"].unsqueeze(1)
        loss = self.mean_masked_loss(batch_dict['pred_sem_residuals'], batch_dict['res_mtrx'], self.reg_loss_func, w,
                                     mask=batch_dict["general_reg_loss_mask"]) * self.occ_fore_res_weight
        return loss, {'occ_loss_res': loss.item()}

    def get_loss(self, batch_dict):
        if self.noloss:
            return torch.tensor(0.0, device="cuda"), {}
        logit = batch_dict['pred_occ_logit']
        if FUSED_LOSS and self.is_softmax and logit.is_cuda and logit.shape[1] == 2 and (not self.reg or self.res_num_dim == 3):
            lw = self.model_cfg.OCC_DENSE_HEAD.LOSS_CONFIG.LOSS_WEIGHTS
            reg = self.reg
            total, out = OccLossFunction.apply(
                logit, batch_dict['pred_sem_residuals'] if reg else None, batch_dict['res_mtrx'] if reg else None,
                batch_dict["pos_mask"], batch_dict['general_cls_loss_mask'], batch_dict["general_cls_loss_mask_float"],
                batch_dict["general_reg_loss_mask"] if reg else None, batch_dict["general_reg_loss_mask_float"] if reg else None,
                lw['res_beta'], self.occ_fore_cls_weight, self.occ_fore_res_weight)
            # the scalars the reference logs with .item() (occ_head_template.py:163,171): one asynchronous copy to pinned
            # memory for both; the wait happens when a value is READ (float(), formatting, arithmetic), not here -- an .item()
            # at this point stalls the host until the whole forward pass has drained
            vals = _lazy_scalars(out.detach(), 2 if reg else 1)
            tb_dict = {'occ_loss_cls': vals[0]}
            if reg:
                tb_dict['occ_loss_res'] = vals[1]
            return total, tb_dict      # (without REG the regression part is an exact 0)
        occ_loss, tb_dict = self.get_cls_layer_loss(batch_dict)
        if self.reg:
            reg_loss, tb_res = self.get_res_layer_loss(batch_dict)
            occ_loss = occ_loss + reg_loss
            tb_dict.update(tb_res)
        return occ_loss, tb_dict


class OccHead3D(OccHeadTemplate):
    def __init__(self, model_cfg, data_cfg, input_channels, num_class, grid_size):
        lc = model_cfg.OCC_DENSE_HEAD.LOSS_CONFIG
        self.is_softmax = lc.get("CLS_LOSS_TYPE", None) == "softmax"
        super().__init__(model_cfg=model_cfg, data_cfg=data_cfg, num_class=num_class, grid_size=grid_size)
        self.stride = int(model_cfg.BACKBONE_3D.STRIDE)
        upd = model_cfg.get("OCC_PNT_UPDATE", None)
        self.prob_needs_grad = bool(upd.get("PASS_GRAD", False)) if upd is not None else True   # (does PassOccVox propagate into the probability)
        cls_channel = num_class + 1 if self.is_softmax else num_class
        self.conv_cls = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, (self.stride ** 3) * cls_channel, 3, padding=1, bias=True, indice_key='cls_ind'))
        self.logit2prob = torch.nn.Softmax(dim=1) if self.is_softmax else torch.nn.Sigmoid()
        if self.reg:
            self.conv_res = spconv.SparseSequential(
                spconv.SubMConv3d(input_channels, (self.stride ** 3) * self.num_class * self.res_num_dim, 3, padding=1,
                                  bias=False, indice_key='res_ind'))

    def _merge_ok(self):
        return MERGE_HEADS and self.reg and len(self.conv_cls) == 1 and len(self.conv_res) == 1 \
            and self.conv_cls[0].kernel_size == self.conv_res[0].kernel_size and self.conv_cls[0].dilation == self.conv_res[0].dilation

    def premerge(self):
        """build the concatenated head weight EARLY in the forward pass (BtcHotPath.forward calls this before the occupancy
        backbone): autograd runs nodes latest-created first, so the CatBackward that splits dW runs after the whole backbone's
        backward, and the head's weight gradient (150 us at 210 K rows) can stay on the side stream until then
        (ops.set_defer_wgrad_join); the hook below is the join in front of that consumer."""
        if not (self._merge_ok() and self.conv_cls[0].weight.is_cuda):
            return
        from .spconv import ops
        cls, res = self.conv_cls[0], self.conv_res[0]
        w = torch.cat([cls.weight, res.weight], dim=-1)
        zeros = self.__dict__.get("_bias_zeros")     # (a missing bias: zeros made once, not a fill launch per step)
        if zeros is None or zeros.device != cls.weight.device:
            zeros = self.__dict__["_bias_zeros"] = cls.weight.new_zeros(max(cls.out_channels, res.out_channels))
        bias = torch.cat([cls.bias if cls.bias is not None else zeros[:cls.out_channels],
                          res.bias if res.bias is not None else zeros[:res.out_channels]])
        if w.requires_grad and torch.is_grad_enabled():
            w.register_hook(_join_wgrad_hook)
            w._btc_join_before_use = True  # SparseConvFunction: this non-leaf weight's gradient may be deferred
        self._merged = (w, bias)

    def _head_rulebook(self, x):
        """the one submanifold rulebook conv_cls and conv_res share (geometry cache of x.indice_dict)"""
        from .spconv import ops
        cls, res = self.conv_cls[0], self.conv_res[0]
        geom = x.indice_dict.setdefault("__geometry_cache__", {})
        gkey = (x.indices.data_ptr(), tuple(x.indices.shape), tuple(int(v) for v in x.spatial_shape), tuple(cls.kernel_size),
                tuple(cls.dilation), True, False)
        hit = geom.get(gkey, None)
        if hit is None:
            rb = ops.build_rulebook(x.indices, x.batch_size, x.spatial_shape, cls.kernel_size, 1, 0, cls.dilation, 0, True, False)
            geom[gkey] = (rb, x.indices)
        else:
            rb = hit[0]
        for m in (cls, res):
            if m.indice_key is not None:
                x.indice_dict[m.indice_key] = rb
        return rb

    def forward_geometry(self, x):
        """the rulebook(s) forward will need on the backbone's output active set, without the feature kernels"""
        if self._merge_ok() and x.indices.is_cuda:
            self._head_rulebook(x)
        else:
            self.conv_cls.forward_geometry(x)
            if self.reg:
                self.conv_res.forward_geometry(x)

    def _merged_heads(self, x):
        """conv_cls and conv_res see the same tensor with the same geometry: run them as ONE sparse conv with the two
        weight tensors concatenated along Cout (parameters / state_dict untouched; autograd splits the gradient back)"""
        cls, res = self.conv_cls[0], self.conv_res[0]
        if getattr(self, "_merged", None) is None:
            self.premerge()
        (w, bias), self._merged = self._merged, None
        from .spconv import ops
        rb = self._head_rulebook(x)
        out = ops.indice_conv(x.features, w, bias, rb, keep_fp32=True)
        # -> the two dense maps (logits (B, nc, D, H, W), residuals) straight from the merged rows: one fill + one scatter launch
        # (ops.dense_split) where slicing, copying and densifying the two parts took six -- same values
        if out.dtype != torch.float32:     # (a bf16 result: only if the head's channel counts ever allow the bf16 kernels; the dense maps are fp32)
            out = out.float()
        return ops.dense_split(out, x.indices, x.batch_size, x.spatial_shape, cls.out_channels)

    def forward(self, data_dict):
        data_dict = self.prepare_loss_map(data_dict)
        x = data_dict['encoded_spconv_tensor']
        if self._merge_ok() and x.features.is_cuda:
            logit, residuals = self._merged_heads(x)
            data_dict['pred_occ_logit'] = logit
            mask = data_dict["general_cls_loss_mask"]
            if self.is_softmax and logit.shape[1] == 2 and logit.dtype == torch.float32 and mask.dtype == torch.uint8 and mask.is_contiguous() \
                    and not (self.prob_needs_grad and torch.is_grad_enabled() and logit.requires_grad):
                # softmax + slice + mask product in one launch (btc_occ_prob) where the probability is a DETACHED input of PassOccVox
                # (OCC_PNT_UPDATE.PASS_GRAD False, the shipped configuration); with PASS_GRAD the torch ops below keep the graph
                from ._lib import check, lib, ptr, stream_ptr
                lg = logit.detach().contiguous()
                prob = torch.empty((lg.shape[0],) + tuple(lg.shape[2:]), dtype=torch.float32, device=lg.device)
                check(lib().btc_occ_prob(ptr(lg), ptr(mask), int(lg.shape[0]), int(prob[0].numel()), ptr(prob), stream_ptr()), "btc_occ_prob")
                data_dict['batch_pred_occ_prob'] = prob
            else:
                prob = self.logit2prob(logit)[:, -1:, ...]
                data_dict['batch_pred_occ_prob'] = prob[:, -1, ...] * mask
            data_dict['pred_sem_residuals'] = residuals
            return data_dict
        logit = self.conv_cls(x).dense()
        prob = self.logit2prob(logit)[:, -1:, ...]
        data_dict['pred_occ_logit'] = logit
        # inactive cells densify to logit 0 => p = 0.5 (App. D.3): kept, they pass OCC_THRESH inside the loss mask
        data_dict['batch_pred_occ_prob'] = prob[:, -1, ...] * data_dict["general_cls_loss_mask"]
        if self.reg:
            data_dict['pred_sem_residuals'] = self.conv_res(x).dense()
        return data_dict


__all__ = {'OccHeadTemplate': OccHeadTemplate, 'OccHead3D': OccHead3D}
