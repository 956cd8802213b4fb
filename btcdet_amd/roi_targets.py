"""Training targets and losses of the ROI head (behind SURVEY.md §8f rows 1 / 2): what RoIHeadTemplate.assign_targets /
get_loss do around ConvHead (/root/reference/btcdet/models/roi_heads/roi_head_template.py:102-232,
target_assigner/proposal_target_layer.py:8-228, utils/loss_utils.py:309-332, utils/box_utils.py:27-52).

Resident, fixed-shape and free of host synchronisation: the reference walks the scenes in Python, trims the ground-truth list with
`.sum() == 0` probes, branches on `numel()` of every candidate set and draws its samples with numpy / torch CPU generators (three
to five device-to-host read-backs per scene).  Here the IoU matching, the three candidate sets, the quota arithmetic and the draws
are device tensors of fixed size (ROI_PER_IMAGE slots per scene), the branches of the reference are `torch.where`s on 0-d count
tensors, and one `torch.Generator` on the device supplies the uniforms -- the same sampling distribution (foreground without
replacement up to its quota, hard / easy background with replacement in the configured ratio, the degenerate cases as the
reference handles them), not the same random stream.  Deterministic parts (matching, labels, canonical transform, the three
losses) are checked against the reference's own functions (tests/test_hip_roi_targets.py)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import iou3d_nms
from .conv_head import rotate_z
from .dense_head import smooth_l1


def match_rois(rois, roi_labels, gt_boxes):
    """per scene: the best same-class ground-truth box of every roi (proposal_target_layer.py:200-228, the configured
    SAMPLE_ROI_BY_EACH_CLASS).  rois (N, 7+), roi_labels (N,) 1-based, gt_boxes (G, 7 + 1) zero-padded, class in the last column.
    -> (max_overlaps (N,), gt_assignment (N,)); a roi whose class has no box gets overlap 0 and box 0, as the reference leaves it"""
    iou = iou3d_nms.boxes_iou3d_gpu(rois[:, 0:7].contiguous(), gt_boxes[:, 0:7].contiguous())       # zero boxes: IoU 0
    same = roi_labels.view(-1, 1) == gt_boxes[:, -1].long().view(1, -1)
    best, arg = torch.where(same, iou, iou.new_full((), -1.0)).max(dim=1)
    none = best < 0
    return torch.where(none, torch.zeros_like(best), best), torch.where(none, torch.zeros_like(arg), arg)


def _kth_of(mask):
    """indices of the set bits of mask (N,) in ascending order, padded with the unset ones: element j < mask.sum() is the j-th member"""
    return torch.sort((~mask).to(torch.uint8), stable=True)[1]


def sample_rois(max_overlaps, cfg, generator=None):
    """proposal_target_layer.py:117-197 for one scene, on the device: -> ROI_PER_IMAGE sampled roi indices (foreground slots first)."""
    R = int(cfg.ROI_PER_IMAGE)
    fg_quota = int(round(float(cfg.FG_RATIO) * R))
    ov, dev = max_overlaps, max_overlaps.device
    fg = ov >= min(cfg.REG_FG_THRESH, cfg.CLS_FG_THRESH)
    easy = ov < cfg.CLS_BG_THRESH_LO
    hard = (ov < cfg.REG_FG_THRESH) & (ov >= cfg.CLS_BG_THRESH_LO)
    n_fg, n_easy, n_hard = fg.sum(), easy.sum(), hard.sum()
    n_bg = n_easy + n_hard
    u = torch.rand((4, max(R, ov.shape[0])), device=dev, generator=generator)
    # foreground candidates in random order (a random permutation's prefix = sampling without replacement)
    fg_order = torch.sort(torch.where(fg, u[0, :ov.shape[0]], u.new_full((), 2.0)))[1]
    hard_list, easy_list = _kth_of(hard), _kth_of(easy)
    # quotas: fg up to its share when there is background, every slot when there is none, no slot when there is no foreground
    fg_this = torch.where(n_bg > 0, torch.clamp(n_fg, max=fg_quota), torch.where(n_fg > 0, torch.full_like(n_fg, R), torch.zeros_like(n_fg)))
    m = R - fg_this                                                                    # background slots
    hard_num = torch.where((n_hard > 0) & (n_easy > 0), torch.minimum((m.double() * float(cfg.HARD_BG_RATIO)).floor().long(), n_hard),
                           torch.where(n_hard > 0, m, torch.zeros_like(m)))
    j = torch.arange(R, device=dev)
    draw = lambda row, n: torch.clamp((u[row, :R].double() * n.double()).floor().long(), max=torch.clamp(n, min=1) - 1)
    fg_idx = torch.where(n_bg > 0, fg_order[torch.clamp(j, max=ov.shape[0] - 1)], fg_order[draw(1, n_fg)])   # (no background: with replacement)
    bg_idx = torch.where((j - fg_this) < hard_num, hard_list[draw(2, n_hard)], easy_list[draw(3, n_easy)])
    return torch.where(j < fg_this, fg_idx, bg_idx)


class ProposalTargetLayer(nn.Module):
    """rois (B, N, 7+), roi_scores, roi_labels, gt_boxes (B, G, 8) -> the reference's targets_dict (proposal_target_layer.py:13-66)"""

    def __init__(self, roi_sampler_cfg, seed=0):
        super().__init__()
        self.roi_sampler_cfg = roi_sampler_cfg
        self.seed = seed
        self._gen = None

    def _generator(self, device):
        if self._gen is None or self._gen.device != device:
            self._gen = torch.Generator(device=device)
            self._gen.manual_seed(self.seed)
        return self._gen

    @torch.no_grad()
    def forward(self, batch_dict, sampled_inds=None):
        """sampled_inds (B, ROI_PER_IMAGE): use these roi indices instead of drawing (tests: the reference's own draw)"""
        cfg = self.roi_sampler_cfg
        rois, scores, labels, gts = batch_dict["rois"], batch_dict["roi_scores"], batch_dict["roi_labels"], batch_dict["gt_boxes"]
        B = batch_dict["batch_size"]
        out = {k: [] for k in ("rois", "gt_of_rois", "gt_iou_of_rois", "roi_scores", "roi_labels")}
        for b in range(B):
            if cfg.get("SAMPLE_ROI_BY_EACH_CLASS", False):
                ov, assign = match_rois(rois[b], labels[b], gts[b])
            else:
                ov, assign = iou3d_nms.boxes_iou3d_gpu(rois[b][:, 0:7].contiguous(), gts[b][:, 0:7].contiguous()).max(dim=1)
            sel = sample_rois(ov, cfg, self._generator(rois.device) if rois.is_cuda else None) if sampled_inds is None else sampled_inds[b]
            out["rois"].append(rois[b][sel])
            out["gt_of_rois"].append(gts[b][assign[sel]])
            out["gt_iou_of_rois"].append(ov[sel])
            out["roi_scores"].append(scores[b][sel])
            out["roi_labels"].append(labels[b][sel])
        t = {k: torch.stack(v) for k, v in out.items()}
        iou = t["gt_iou_of_rois"]
        t["reg_valid_mask"] = (iou > cfg.REG_FG_THRESH).long()
        if cfg.CLS_SCORE_TYPE == "cls":
            cls = (iou > cfg.CLS_FG_THRESH).long()
            cls[(iou > cfg.CLS_BG_THRESH) & (iou < cfg.CLS_FG_THRESH)] = -1
        elif cfg.CLS_SCORE_TYPE == "roi_iou":
            lo, hi = cfg.CLS_BG_THRESH, cfg.CLS_FG_THRESH
            cls = torch.where(iou > hi, torch.ones_like(iou), torch.where(iou < lo, torch.zeros_like(iou), (iou - lo) / (hi - lo)))
        else:
            raise NotImplementedError(cfg.CLS_SCORE_TYPE)
        t["rcnn_cls_labels"] = cls
        return t


def canonical_targets(targets):
    """roi_head_template.py:102-134: the matched boxes in their roi's frame (centre at the origin, heading 0), heading folded into
    [-pi/2, pi/2]; keeps the untransformed copy as gt_of_rois_src"""
    rois, gt = targets["rois"], targets["gt_of_rois"]
    B = rois.shape[0]
    targets["gt_of_rois_src"] = gt.clone().detach()
    ry = rois[:, :, 6] % (2 * math.pi)
    local = torch.cat([gt[:, :, 0:3] - rois[:, :, 0:3], gt[:, :, 3:6], (gt[:, :, 6] - ry).unsqueeze(-1), gt[:, :, 7:]], dim=-1)
    xyz = rotate_z(local.view(-1, 1, local.shape[-1])[:, :, 0:3], -ry.view(-1)).view(B, -1, 3)
    heading = local[:, :, 6] % (2 * math.pi)
    opposite = (heading > math.pi * 0.5) & (heading < math.pi * 1.5)
    heading = torch.where(opposite, (heading + math.pi) % (2 * math.pi), heading)
    heading = torch.where(heading > math.pi, heading - 2 * math.pi, heading)
    heading = torch.clamp(heading, min=-math.pi / 2, max=math.pi / 2)
    targets["gt_of_rois"] = torch.cat([xyz, local[:, :, 3:6], heading.unsqueeze(-1), local[:, :, 7:]], dim=-1)
    return targets


def boxes_to_corners_3d(boxes):
    """(N, 7) -> (N, 8, 3), the reference's corner order (box_utils.py:27-52)"""
    t = boxes.new_tensor([[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1], [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]]) / 2
    corners = boxes[:, None, 3:6] * t[None]
    return rotate_z(corners, boxes[:, 6]) + boxes[:, None, 0:3]


def corner_loss_lidar(pred, gt):
    """(N, 7) x 2 -> (N,): smooth-L1 (beta 1) of the corner distances, against the box or its half-turn twin, whichever is nearer"""
    pc = boxes_to_corners_3d(pred)
    flip = torch.cat([gt[:, 0:6], gt[:, 6:7] + math.pi], dim=-1)
    dist = torch.min(torch.norm(pc - boxes_to_corners_3d(gt), dim=2), torch.norm(pc - boxes_to_corners_3d(flip), dim=2))
    return torch.where(dist < 1.0, 0.5 * dist * dist, dist - 0.5).mean(dim=1)


def rcnn_cls_loss(rcnn_cls, labels, loss_cfg):
    """roi_head_template.py:203-220"""
    labels = labels.view(-1)
    if loss_cfg.CLS_LOSS == "BinaryCrossEntropy":
        per = F.binary_cross_entropy(torch.sigmoid(rcnn_cls.view(-1)), labels.float().clamp(min=0), reduction="none")
    elif loss_cfg.CLS_LOSS == "CrossEntropy":
        per = F.cross_entropy(rcnn_cls, labels, reduction="none", ignore_index=-1)
    else:
        raise NotImplementedError(loss_cfg.CLS_LOSS)
    valid = (labels >= 0).float()
    return (per * valid).sum() / torch.clamp(valid.sum(), min=1.0) * loss_cfg.LOSS_WEIGHTS["rcnn_cls_weight"]


def rcnn_reg_loss(ret, box_coder, loss_cfg):
    """roi_head_template.py:136-201 without its two `.item()` read-backs: the foreground mean is a masked sum over all sampled rois
    (identical for the foreground rows, zero when there is none -- the reference's `fg_sum > 0` branch).  -> (loss, corner term)"""
    if loss_cfg.REG_LOSS != "smooth-l1":
        raise NotImplementedError(loss_cfg.REG_LOSS)
    code = box_coder.code_size
    fg = (ret["reg_valid_mask"].view(-1) > 0).float()
    n_fg = torch.clamp(fg.sum(), min=1.0)
    gt_ct = ret["gt_of_rois"][..., 0:code].reshape(-1, code)
    rois = ret["rois"].reshape(-1, ret["rois"].shape[-1])[:, 0:code]
    reg = ret["rcnn_reg"].view(gt_ct.shape[0], -1)
    anchor = rois.clone().detach()
    anchor[:, 0:3] = 0
    anchor[:, 6] = 0
    target = box_coder.encode_torch(gt_ct.clone(), anchor)
    # (code_weights are configured but never applied by the reference's WeightedSmoothL1Loss: loss_utils.py:224-227)
    loss = smooth_l1(reg.unsqueeze(0), target.unsqueeze(0), fg.unsqueeze(0), beta=1.0 / 9.0).sum() / n_fg
    loss = loss * loss_cfg.LOSS_WEIGHTS["rcnn_reg_weight"]
    corner = None
    if loss_cfg.CORNER_LOSS_REGULARIZATION:
        anchors = rois.clone().detach().unsqueeze(0)
        anchors[:, :, 0:3] = 0
        boxes = box_coder.decode_torch(reg.view(1, -1, code), anchors).view(-1, code)
        xyz = rotate_z(boxes[:, None, 0:3], rois[:, 6])[:, 0] + rois[:, 0:3]
        boxes = torch.cat([xyz, boxes[:, 3:7]], dim=-1)                      # (the decoded heading already carries the roi's)
        per = corner_loss_lidar(boxes[:, 0:7], ret["gt_of_rois_src"][..., 0:7].reshape(-1, 7))
        corner = (per * fg).sum() / n_fg * loss_cfg.LOSS_WEIGHTS["rcnn_corner_weight"]
        loss = loss + corner
    return loss, corner
