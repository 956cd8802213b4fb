"""Anchor-based dense head on the BEV map + the proposal step that feeds the ROI head (the glue around the
rotated IoU / NMS operator of csrc/iou3d_nms.hip).

Same module protocol, parameter names (`conv_cls`, `conv_box`, `conv_dir_cls`), batch_dict keys and arithmetic as
  AnchorHeadSingle / AnchorHeadTemplate   /root/reference/btcdet/models/dense_heads/anchor_head_single.py:7-82, anchor_head_template.py:11-277
  AnchorGenerator                         dense_heads/target_assigner/anchor_generator.py:4-60
  AxisAlignedTargetAssigner               dense_heads/target_assigner/axis_aligned_target_assigner.py:8-213 (single head, POS_FRACTION < 0 and >= 0)
  ResidualCoder                           btcdet/utils/box_coder_utils.py:78-151
  class_agnostic_nms, proposal_layer      models/model_utils/model_nms_utils.py:6-25, roi_heads/roi_head_template.py:45-100
checked against those modules run here (tests/golden/gen_head_golden.py -> tests/test_hip_dense_head.py).  Dense 1x1 convs are
the vendor library's; the NMS inside the proposal step is this repository's HIP kernel (btcdet_amd.iou3d_nms.nms_gpu)."""
import os

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import iou3d_nms


def limit_period(val, offset=0.5, period=np.pi):
    return val - torch.floor(val / period + offset) * period


# ------------------------------------------------------------------------------------------------ anchors and box coding
def make_anchors(point_cloud_range, generator_cfg, grid_size, device):
    """per class config: anchors [nz, ny, nx, n_sizes, n_rotations, 7] on the stride-reduced BEV grid, z at the box centre;
    -> (list of tensors, anchors per location per config)"""
    out, per_loc = [], []
    r = [float(v) for v in point_cloud_range]
    for cfg in generator_cfg:
        fx, fy = (int(v) for v in (np.asarray(grid_size[:2]) // cfg["feature_map_stride"]))
        sizes, rots, heights = cfg["anchor_sizes"], cfg["anchor_rotations"], cfg["anchor_bottom_heights"]
        per_loc.append(len(rots) * len(sizes) * len(heights))
        if cfg.get("align_center", False):
            sx, sy = (r[3] - r[0]) / fx, (r[4] - r[1]) / fy
            ox, oy = sx / 2, sy / 2
        else:
            sx, sy = (r[3] - r[0]) / (fx - 1), (r[4] - r[1]) / (fy - 1)
            ox, oy = 0, 0
        xs = torch.arange(r[0] + ox, r[3] + 1e-5, step=sx, dtype=torch.float32)
        ys = torch.arange(r[1] + oy, r[4] + 1e-5, step=sy, dtype=torch.float32)
        zs = torch.tensor(heights, dtype=torch.float32)
        gz, gy, gx = torch.meshgrid(zs, ys, xs, indexing="ij")
        centre = torch.stack((gx, gy, gz), dim=-1)[:, :, :, None, None, :]                       # [nz, ny, nx, 1, 1, 3]
        size = torch.tensor(sizes, dtype=torch.float32).view(1, 1, 1, -1, 1, 3)
        rot = torch.tensor(rots, dtype=torch.float32).view(1, 1, 1, 1, -1, 1)
        shape = (len(heights), ys.numel(), xs.numel(), len(sizes), len(rots))
        a = torch.cat((centre.expand(*shape, 3), size.expand(*shape, 3), rot.expand(*shape, 1)), dim=-1).contiguous()
        a[..., 2] += a[..., 5] / 2
        out.append(a.to(device))
    return out, per_loc


class ResidualCoder(object):
    """(x, y) offsets in units of the anchor's BEV diagonal, z in units of its height, log size ratios, heading difference"""

    def __init__(self, code_size=7, encode_angle_by_sincos=False, **kwargs):
        self.code_size = code_size + (1 if encode_angle_by_sincos else 0)
        self.encode_angle_by_sincos = encode_angle_by_sincos

    def encode_torch(self, boxes, anchors):
        anchors[:, 3:6] = torch.clamp_min(anchors[:, 3:6], min=1e-5)     # in place, as the reference does
        boxes[:, 3:6] = torch.clamp_min(boxes[:, 3:6], min=1e-5)
        a, g = anchors, boxes
        diag = torch.sqrt(a[:, 3:4] ** 2 + a[:, 4:5] ** 2)
        parts = [(g[:, 0:1] - a[:, 0:1]) / diag, (g[:, 1:2] - a[:, 1:2]) / diag, (g[:, 2:3] - a[:, 2:3]) / a[:, 5:6],
                 torch.log(g[:, 3:4] / a[:, 3:4]), torch.log(g[:, 4:5] / a[:, 4:5]), torch.log(g[:, 5:6] / a[:, 5:6])]
        if self.encode_angle_by_sincos:
            parts += [torch.cos(g[:, 6:7]) - torch.cos(a[:, 6:7]), torch.sin(g[:, 6:7]) - torch.sin(a[:, 6:7])]
        else:
            parts.append(g[:, 6:7] - a[:, 6:7])
        parts.append(g[:, 7:] - a[:, 7:])
        return torch.cat(parts, dim=-1)

    def decode_torch(self, enc, anchors):
        a = anchors
        diag = torch.sqrt(a[..., 3:4] ** 2 + a[..., 4:5] ** 2)
        parts = [enc[..., 0:1] * diag + a[..., 0:1], enc[..., 1:2] * diag + a[..., 1:2], enc[..., 2:3] * a[..., 5:6] + a[..., 2:3],
                 torch.exp(enc[..., 3:4]) * a[..., 3:4], torch.exp(enc[..., 4:5]) * a[..., 4:5], torch.exp(enc[..., 5:6]) * a[..., 5:6]]
        if self.encode_angle_by_sincos:
            parts.append(torch.atan2(enc[..., 7:8] + torch.sin(a[..., 6:7]), enc[..., 6:7] + torch.cos(a[..., 6:7])))
            rest = enc[..., 8:] + a[..., 7:]
        else:
            parts.append(enc[..., 6:7] + a[..., 6:7])
            rest = enc[..., 7:] + a[..., 7:]
        parts.append(rest)
        return torch.cat(parts, dim=-1)


def nearest_bev_iou(boxes_a, boxes_b):
    """IoU of the axis-aligned BEV footprints after snapping every heading to the nearer axis (box_utils.boxes3d_nearest_bev_iou)"""
    def aligned(b):
        rot = limit_period(b[:, 6], 0.5, np.pi).abs()
        dims = torch.where(rot[:, None] < np.pi / 4, b[:, [3, 4]], b[:, [4, 3]])
        return torch.cat((b[:, 0:2] - dims / 2, b[:, 0:2] + dims / 2), dim=1)
    a, b = aligned(boxes_a), aligned(boxes_b)
    w = torch.clamp_min(torch.min(a[:, 2, None], b[None, :, 2]) - torch.max(a[:, 0, None], b[None, :, 0]), min=0)
    h = torch.clamp_min(torch.min(a[:, 3, None], b[None, :, 3]) - torch.max(a[:, 1, None], b[None, :, 1]), min=0)
    inter = w * h
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / torch.clamp_min(area_a[:, None] + area_b[None, :] - inter, min=1e-6)


# ------------------------------------------------------------------------------------------------ target assignment
class AxisAlignedTargetAssigner(object):
    def __init__(self, model_cfg, class_names, box_coder, match_height=False):
        gen, tgt = model_cfg.ANCHOR_GENERATOR_CONFIG, model_cfg.TARGET_ASSIGNER_CONFIG
        self.box_coder, self.match_height = box_coder, match_height
        self.class_names = np.array(class_names)
        self.anchor_class_names = [c["class_name"] for c in gen]
        self.pos_fraction = tgt.POS_FRACTION if tgt.POS_FRACTION >= 0 else None
        self.sample_size, self.norm_by_num_examples = tgt.SAMPLE_SIZE, tgt.NORM_BY_NUM_EXAMPLES
        self.matched = {c["class_name"]: c["matched_threshold"] for c in gen}
        self.unmatched = {c["class_name"]: c["unmatched_threshold"] for c in gen}
        if model_cfg.get("USE_MULTIHEAD", False):
            raise NotImplementedError("multi-head anchor assignment is not on the configured path")

    def assign_targets(self, all_anchors, gt_boxes_with_classes):
        """all_anchors: list per class config of [nz, ny, nx, n_size, n_rot, code]; gt (B, M, code + 1), zero rows pad the tail
        -> box_cls_labels (B, A) int32 (-1 ignore, 0 background, class id), box_reg_targets (B, A, code), reg_weights (B, A)"""
        labels, targets, weights = [], [], []
        for gt in gt_boxes_with_classes:
            n = gt.shape[0]
            while n > 1 and gt[n - 1, :-1].sum() == 0:      # the reference keeps row 0 even when it is padding
                n -= 1
            boxes, classes = gt[:n, :-1], gt[:n, -1].int()
            per_cfg = []
            for name, anchors in zip(self.anchor_class_names, all_anchors):
                of_class = torch.from_numpy(np.atleast_1d(self.class_names[classes.cpu().numpy() - 1] == name)).to(gt.device)
                grid = anchors.shape[:3]
                one = self._assign_single(anchors.reshape(-1, anchors.shape[-1]), boxes[of_class], classes[of_class], self.matched[name],
                                          self.unmatched[name])
                per_cfg.append((one[0].view(*grid, -1), one[1].view(*grid, -1, self.box_coder.code_size), one[2].view(*grid, -1)))
            labels.append(torch.cat([c[0] for c in per_cfg], dim=-1).view(-1))
            targets.append(torch.cat([c[1] for c in per_cfg], dim=-2).view(-1, self.box_coder.code_size))
            weights.append(torch.cat([c[2] for c in per_cfg], dim=-1).view(-1))
        return {"box_cls_labels": torch.stack(labels), "box_reg_targets": torch.stack(targets), "reg_weights": torch.stack(weights)}

    def _assign_single(self, anchors, gt_boxes, gt_classes, matched_threshold, unmatched_threshold):
        A, G, dev = anchors.shape[0], gt_boxes.shape[0], anchors.device
        labels = torch.full((A,), -1, dtype=torch.int32, device=dev)
        best_gt = None
        if G > 0 and A > 0:
            iou = iou3d_nms.boxes_iou3d_gpu(anchors[:, 0:7], gt_boxes[:, 0:7]) if self.match_height else nearest_bev_iou(anchors[:, 0:7], gt_boxes[:, 0:7])
            best_gt = iou.argmax(dim=1)
            best_iou = iou.gather(1, best_gt[:, None])[:, 0]
            top_per_gt = iou.max(dim=0).values
            top_per_gt[top_per_gt == 0] = -1                           # a ground-truth box no anchor touches forces nothing
            forced = torch.nonzero(iou == top_per_gt)[:, 0]            # every anchor that ties a box's best overlap
            forced_cls = gt_classes[best_gt[forced]]
            labels[forced] = forced_cls
            pos = best_iou >= matched_threshold
            labels[pos] = gt_classes[best_gt[pos]]
            bg = torch.nonzero(best_iou < unmatched_threshold)[:, 0]
        else:
            bg = torch.arange(A, device=dev)
        fg = torch.nonzero(labels > 0)[:, 0]
        if self.pos_fraction is not None:
            cap = int(self.pos_fraction * self.sample_size)
            if len(fg) > cap:
                labels[torch.randperm(len(fg))[:len(fg) - cap]] = -1   # indexes `labels` directly, as the reference does
                fg = torch.nonzero(labels > 0)[:, 0]
            n_bg = self.sample_size - int((labels > 0).sum())
            if len(bg) > n_bg:
                labels[bg[torch.randint(0, len(bg), size=(n_bg,))]] = 0
        elif G == 0 or A == 0:
            labels[:] = 0
        else:
            labels[bg] = 0
            labels[forced] = forced_cls
        targets = anchors.new_zeros((A, self.box_coder.code_size))
        if G > 0 and A > 0:
            targets[fg] = self.box_coder.encode_torch(gt_boxes[best_gt[fg], :], anchors[fg, :])
        weights = anchors.new_zeros((A,))
        if self.norm_by_num_examples:
            weights[labels > 0] = 1.0 / max(float((labels >= 0).sum()), 1.0)
        else:
            weights[labels > 0] = 1.0
        return labels, targets, weights


# ------------------------------------------------------------------------------------------------ losses
def sigmoid_focal_loss(logits, one_hot, weights, alpha=0.25, gamma=2.0):
    p = torch.sigmoid(logits)
    a = one_hot * alpha + (1 - one_hot) * (1 - alpha)
    pt = one_hot * (1.0 - p) + (1.0 - one_hot) * p
    bce = torch.clamp(logits, min=0) - logits * one_hot + torch.log1p(torch.exp(-torch.abs(logits)))
    return a * torch.pow(pt, gamma) * bce * weights.unsqueeze(-1)


def smooth_l1(pred, target, weights, beta=1.0 / 9.0):
    target = torch.where(torch.isnan(target), pred, target)
    n = torch.abs(pred - target)
    loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta) if beta >= 1e-5 else n
    return loss * weights.unsqueeze(-1)        # (the reference builds code_weights but never applies them: loss_utils.py:224-227)


# ------------------------------------------------------------------------------------------------ the head
class AnchorHeadSingle(nn.Module):
    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range, predict_boxes_when_training=True):
        super().__init__()
        self.model_cfg, self.num_class, self.class_names = model_cfg, num_class, class_names
        self.predict_boxes_when_training = predict_boxes_when_training
        tgt = model_cfg.TARGET_ASSIGNER_CONFIG
        assert tgt.BOX_CODER == "ResidualCoder" and tgt.NAME == "AxisAlignedTargetAssigner", "the configured coder / assigner only"
        self.box_coder = ResidualCoder(num_dir_bins=tgt.get("NUM_DIR_BINS", 6), **tgt.get("BOX_CODER_CONFIG", {}))
        self.point_cloud_range, self.grid_size = point_cloud_range, grid_size
        self._anchors = {}
        _, per_loc = make_anchors(point_cloud_range, model_cfg.ANCHOR_GENERATOR_CONFIG, grid_size, "cpu")
        self.num_anchors_per_location = sum(per_loc)
        self.target_assigner = AxisAlignedTargetAssigner(model_cfg, class_names, self.box_coder, tgt.MATCH_HEIGHT)
        self.forward_ret_dict = {}
        A = self.num_anchors_per_location
        self.conv_cls = nn.Conv2d(input_channels, A * num_class, kernel_size=1)
        self.conv_box = nn.Conv2d(input_channels, A * self.box_coder.code_size, kernel_size=1)
        self.conv_dir_cls = None
        if model_cfg.get("USE_DIRECTION_CLASSIFIER", None) is not None:
            self.conv_dir_cls = nn.Conv2d(input_channels, A * model_cfg.NUM_DIR_BINS, kernel_size=1)
        nn.init.constant_(self.conv_cls.bias, -np.log((1 - 0.01) / 0.01))
        nn.init.normal_(self.conv_box.weight, mean=0, std=0.001)

    def anchors(self, device):
        key = str(device)
        if key not in self._anchors:
            self._anchors[key] = make_anchors(self.point_cloud_range, self.model_cfg.ANCHOR_GENERATOR_CONFIG, self.grid_size, device)[0]
        return self._anchors[key]

    def forward(self, data_dict):
        x = data_dict["spatial_features_2d"]
        cls = self.conv_cls(x).permute(0, 2, 3, 1).contiguous()
        box = self.conv_box(x).permute(0, 2, 3, 1).contiguous()
        dirs = self.conv_dir_cls(x).permute(0, 2, 3, 1).contiguous() if self.conv_dir_cls is not None else None
        self.forward_ret_dict = {"cls_preds": cls, "box_preds": box}
        if dirs is not None:
            self.forward_ret_dict["dir_cls_preds"] = dirs
        if self.training:
            targets = data_dict.pop("rpn_targets", None)   # assigned ahead of the forward pass (BtcHotPath.prepare), else here
            if targets is None:
                targets = self.target_assigner.assign_targets(self.anchors(x.device), data_dict["gt_boxes"])
            self.forward_ret_dict.update(targets)
        if not self.training or self.predict_boxes_when_training:
            data_dict["batch_cls_preds"], data_dict["batch_box_preds"] = self.generate_predicted_boxes(data_dict["batch_size"], cls, box, dirs)
            data_dict["cls_preds_normalized"] = False
        return data_dict

    def _flat_anchors(self, device, batch_size):
        a = torch.cat(self.anchors(device), dim=-3)
        return a.view(1, -1, a.shape[-1]).repeat(batch_size, 1, 1)

    def generate_predicted_boxes(self, batch_size, cls_preds, box_preds, dir_cls_preds=None):
        anchors = self._flat_anchors(cls_preds.device, batch_size)
        n = anchors.shape[1]
        scores = cls_preds.view(batch_size, n, -1).float()
        boxes = self.box_coder.decode_torch(box_preds.view(batch_size, n, -1), anchors)
        if dir_cls_preds is not None:
            off, lim, bins = self.model_cfg.DIR_OFFSET, self.model_cfg.DIR_LIMIT_OFFSET, self.model_cfg.NUM_DIR_BINS
            which = torch.max(dir_cls_preds.view(batch_size, n, -1), dim=-1)[1]
            period = 2 * np.pi / bins
            boxes[..., 6] = limit_period(boxes[..., 6] - off, lim, period) + off + period * which.to(boxes.dtype)
        return scores, boxes

    def get_loss(self):
        f, w = self.forward_ret_dict, self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        cls_preds, labels = f["cls_preds"], f["box_cls_labels"]
        B = int(cls_preds.shape[0])
        positives, cared = labels > 0, labels >= 0
        n_pos = torch.clamp(positives.sum(1, keepdim=True).float(), min=1.0)
        cls_w = ((labels == 0) * 1.0 + 1.0 * positives).float() / n_pos
        reg_w = positives.float() / n_pos
        if self.num_class == 1:
            labels[positives] = 1
        cls_t = (labels * cared.type_as(labels)).long()
        one_hot = torch.zeros(*cls_t.shape, self.num_class + 1, dtype=cls_preds.dtype, device=cls_preds.device).scatter_(-1, cls_t.unsqueeze(-1), 1.0)[..., 1:]
        cls_loss = sigmoid_focal_loss(cls_preds.view(B, -1, self.num_class), one_hot, cls_w).sum() / B * w["cls_weight"]
        parts = [("rpn_loss_cls", cls_loss)]
        anchors = self._flat_anchors(cls_preds.device, B)
        box_preds, reg_t = f["box_preds"].view(B, -1, f["box_preds"].shape[-1] // self.num_anchors_per_location), f["box_reg_targets"]
        # sin(a - b) = sin a cos b - cos a sin b: the heading residual enters as that pair of products
        p_rot, t_rot = torch.sin(box_preds[..., 6:7]) * torch.cos(reg_t[..., 6:7]), torch.cos(box_preds[..., 6:7]) * torch.sin(reg_t[..., 6:7])
        p = torch.cat([box_preds[..., :6], p_rot, box_preds[..., 7:]], dim=-1)
        t = torch.cat([reg_t[..., :6], t_rot, reg_t[..., 7:]], dim=-1)
        loc_loss = smooth_l1(p, t, reg_w).sum() / B * w["loc_weight"]
        box_loss = loc_loss
        parts.append(("rpn_loss_loc", loc_loss))
        if "dir_cls_preds" in f:
            bins = self.model_cfg.NUM_DIR_BINS
            rot_gt = reg_t[..., 6] + anchors[..., 6]
            dir_t = torch.clamp(torch.floor(limit_period(rot_gt - self.model_cfg.DIR_OFFSET, 0, 2 * np.pi) / (2 * np.pi / bins)).long(), min=0, max=bins - 1)
            dw = positives.type_as(cls_preds)
            dw = dw / torch.clamp(dw.sum(-1, keepdim=True), min=1.0)
            dir_loss = (F.cross_entropy(f["dir_cls_preds"].view(B, -1, bins).permute(0, 2, 1), dir_t, reduction="none") * dw).sum() / B * w["dir_weight"]
            box_loss = box_loss + dir_loss
            parts.append(("rpn_loss_dir", dir_loss))
        loss = cls_loss + box_loss
        parts.append(("rpn_loss", loss))
        # the logged scalars (anchor_head_template.py: .item() per term) travel as ONE asynchronous copy and wait when read
        from .occ_head import _lazy_scalars
        vals = _lazy_scalars(torch.stack([v.detach().float() for _, v in parts]), len(parts))
        tb = {k: v for (k, _), v in zip(parts, vals)}
        return loss, tb


# ------------------------------------------------------------------------------------------------ proposals
def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    """-> (indices into the inputs of the kept boxes in descending score order, their scores)"""
    src_scores = box_scores
    if score_thresh is not None:
        above = box_scores >= score_thresh
        box_scores, box_preds = box_scores[above], box_preds[above]
    selected = []
    if box_scores.shape[0] > 0:
        top_scores, top = torch.topk(box_scores, k=min(nms_config.NMS_PRE_MAXSIZE, box_scores.shape[0]))
        keep, _ = getattr(iou3d_nms, nms_config.NMS_TYPE)(box_preds[top][:, 0:7], top_scores, nms_config.NMS_THRESH, **nms_config)
        selected = top[keep[:nms_config.NMS_POST_MAXSIZE]]
    if score_thresh is not None:
        selected = torch.nonzero(above).view(-1)[selected]
    return selected, src_scores[selected]


@torch.no_grad()
def proposal_layer(batch_dict, nms_config):
    """RoIHeadTemplate.proposal_layer: per scene, class-agnostic NMS over the dense head's boxes -> rois (B, NMS_POST_MAXSIZE, code),
    roi_scores, roi_labels (1-based), zero padded"""
    B, boxes, scores = batch_dict["batch_size"], batch_dict["batch_box_preds"], batch_dict["batch_cls_preds"]
    K = nms_config.NMS_POST_MAXSIZE
    rois = boxes.new_zeros((B, K, boxes.shape[-1]))
    roi_scores = boxes.new_zeros((B, K))
    roi_labels = boxes.new_zeros((B, K), dtype=torch.long)
    if nms_config.MULTI_CLASSES_NMS:
        raise NotImplementedError
    if batch_dict.get("batch_index", None) is None and boxes.is_cuda and boxes.dim() == 3 and boxes.shape[1] > 0 and boxes.shape[-1] == 7 \
            and nms_config.NMS_TYPE in ("nms_gpu", "nms_normal_gpu"):
        # every scene at once, resident: top-k by score, the greedy chain stopped at NMS_POST_MAXSIZE kept boxes (iou3d_nms.nms_topk --
        # exactly the reference's keep[:NMS_POST_MAXSIZE], model_nms_utils.py:6-25), padded rows gathered as zeros.  No read-back; the
        # per-scene loop below is the reference's own shape (one full chain over NMS_PRE_MAXSIZE boxes and two read-backs per scene).
        best, label = torch.max(scores, dim=2)                                                     # (B, A)
        top_scores, top = torch.topk(best, k=min(nms_config.NMS_PRE_MAXSIZE, best.shape[1]), dim=1)   # sorted, descending
        cand = torch.gather(boxes, 1, top.unsqueeze(-1).expand(-1, -1, boxes.shape[-1]))
        keep, _ = iou3d_nms.nms_topk(cand[..., 0:7], nms_config.NMS_THRESH, K, rotated=nms_config.NMS_TYPE == "nms_gpu")
        valid = keep >= 0
        sel = torch.gather(top, 1, keep.clamp(min=0))                                              # (B, K) anchor indices
        rois = torch.gather(boxes, 1, sel.unsqueeze(-1).expand(-1, -1, boxes.shape[-1])) * valid.unsqueeze(-1).to(boxes.dtype)
        roi_scores = torch.gather(best, 1, sel) * valid.to(best.dtype)
        roi_labels = torch.gather(label, 1, sel) * valid.to(label.dtype)
        batch_dict.update(rois=rois, roi_scores=roi_scores, roi_labels=roi_labels + 1, has_class_labels=scores.shape[-1] > 1)
        batch_dict.pop("batch_index", None)
        return batch_dict
    for b in range(B):
        pick = (batch_dict["batch_index"] == b) if batch_dict.get("batch_index", None) is not None else b
        bx, sc = boxes[pick], scores[pick]
        best, label = torch.max(sc, dim=1)
        sel, _ = class_agnostic_nms(best, bx, nms_config)
        rois[b, :len(sel)], roi_scores[b, :len(sel)], roi_labels[b, :len(sel)] = bx[sel], best[sel], label[sel]
    batch_dict.update(rois=rois, roi_scores=roi_scores, roi_labels=roi_labels + 1, has_class_labels=scores.shape[-1] > 1)
    batch_dict.pop("batch_index", None)
    return batch_dict


__all__ = {"AnchorHeadSingle": AnchorHeadSingle}
