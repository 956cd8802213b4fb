"""Golden vectors of the ROI head's training targets and losses from the REFERENCE's own RoIHeadTemplate / ProposalTargetLayer
(btcdet/models/roi_heads/roi_head_template.py:102-232, target_assigner/proposal_target_layer.py), executed here on CPU with the one
compiled primitive they call (iou3d_nms_cuda.boxes_overlap_bev_gpu) served by the C oracle's rotated-overlap restatement.

    python tests/golden/gen_roi_targets_golden.py   ->  tests/golden/roi_targets.npz

The reference draws its samples from numpy / torch CPU generators; the drawn roi indices are recorded (subsample_rois is wrapped) so
that the test can hand the SAME draw to btcdet_amd.roi_targets and compare everything that is deterministic given the draw: the IoU
matching, the sampled tensors, labels and masks, the canonical transform, and the three loss terms on hash-valued head outputs."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_env  # noqa: E402
import oracle_spconv  # noqa: E402

ref_env.install(oracle_spconv)
import torch  # noqa: E402

import common  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    ans_overlap.copy_(torch.from_numpy(orc.boxes_overlap_bev(boxes_a.detach().numpy().astype(np.float32), boxes_b.detach().numpy().astype(np.float32))))


sys.modules["btcdet.ops.iou3d_nms.iou3d_nms_cuda"].boxes_overlap_bev_gpu = boxes_overlap_bev_gpu
torch.cuda.FloatTensor = lambda *a: torch.zeros(*[tuple(x) if isinstance(x, torch.Size) else x for x in a], dtype=torch.float32)
from btcdet.models.roi_heads.roi_head_template import RoIHeadTemplate  # noqa: E402
from btcdet.models.roi_heads.target_assigner.proposal_target_layer import ProposalTargetLayer  # noqa: E402


def main():
    cfg = ref_env.load_ref_cfg()
    head = RoIHeadTemplate(num_class=1, model_cfg=cfg.MODEL.ROI_HEAD)
    head.build_losses(cfg.MODEL.ROI_HEAD.LOSS_CONFIG)
    inp = common.roi_target_inputs()
    bd = {k: (torch.from_numpy(v.copy()) if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
    drawn, overlaps = [], []
    orig = ProposalTargetLayer.subsample_rois

    def recording(self, max_overlaps):
        sel = orig(self, max_overlaps)
        drawn.append(sel.numpy().copy())
        overlaps.append(max_overlaps.numpy().copy())
        return sel

    ProposalTargetLayer.subsample_rois = recording
    np.random.seed(7)
    torch.manual_seed(7)
    t = head.assign_targets(bd)
    gold = {"sampled_inds": np.stack(drawn), "max_overlaps": np.stack(overlaps)}
    for k in ("rois", "gt_of_rois", "gt_of_rois_src", "gt_iou_of_rois", "roi_scores", "roi_labels", "reg_valid_mask", "rcnn_cls_labels"):
        gold["t_" + k] = t[k].numpy()
    B, R = t["rois"].shape[:2]
    cls = (common._hash01(B * R, 320) - np.float32(0.5)) * np.float32(4.0)
    reg = (common._hash01(B * R * 7, 321).reshape(B * R, 7) - np.float32(0.5)) * np.float32(0.6)
    head.forward_ret_dict = dict(t, rcnn_cls=torch.from_numpy(cls.reshape(B * R, 1)), rcnn_reg=torch.from_numpy(reg))
    loss, tb = head.get_loss()
    gold["loss"] = np.array([float(loss), tb["rcnn_loss_cls"], tb["rcnn_loss_reg"], tb.get("rcnn_loss_corner", 0.0)], np.float64)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in gold.items()}, gold["loss"],
          "fg per scene", (gold["t_reg_valid_mask"] > 0).sum(1), "max iou", gold["max_overlaps"].max(1))
    np.savez_compressed(os.path.join(HERE, "roi_targets.npz"), **gold)


if __name__ == "__main__":
    main()
