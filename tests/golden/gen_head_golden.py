"""Golden vectors of the reference's BEV backbone + anchor head + proposal step, produced by its OWN modules
run here on CPU: BaseBEVBackbone, AnchorHeadSingle (forward in training mode, get_loss, generate_predicted_boxes with the
AxisAlignedTargetAssigner / ResidualCoder it builds), RoIHeadTemplate.proposal_layer (class_agnostic_nms) -- with the one
compiled primitive they call (iou3d_nms_cuda.nms_gpu) served by the C oracle's NMS.

    python tests/golden/gen_head_golden.py   ->  tests/golden/head.npz"""
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


def nms_gpu(boxes, keep, thresh):
    b = boxes.detach().cpu().numpy().astype(np.float32)
    k = orc.nms(b, -np.arange(b.shape[0], dtype=np.float32), float(thresh), None, True)      # boxes arrive sorted by score
    keep[:len(k)] = torch.from_numpy(np.asarray(k, dtype=np.int64))
    return len(k)


sys.modules["btcdet.ops.iou3d_nms.iou3d_nms_cuda"].nms_gpu = nms_gpu
from btcdet.models.backbones_2d.base_bev_backbone import BaseBEVBackbone  # noqa: E402
from btcdet.models.dense_heads.anchor_head_single import AnchorHeadSingle  # noqa: E402
from btcdet.models.roi_heads.roi_head_template import RoIHeadTemplate  # noqa: E402


def inputs():
    sf = (common._hash01(2 * 256 * 200 * 176, 9) - np.float32(0.35)).clip(0).reshape(2, 256, 200, 176)     # sparse-ish, non-negative like a ReLU map
    from btcdet_amd import synth
    boxes = [synth.make_scene(s, az_step=2.0)["gt_boxes"] for s in (41, 42)]
    g = max(len(b) for b in boxes)
    gt = np.zeros((2, g + 1, 8), np.float32)            # one padding row more than the fuller scene has boxes
    for i, b in enumerate(boxes):
        gt[i, :len(b)] = b
    return sf, gt


if __name__ == "__main__":
    cfg = ref_env.load_ref_cfg()
    m = cfg.MODEL
    sf, gt = inputs()
    gold = {"gt_boxes": gt}
    bev = BaseBEVBackbone(m.BACKBONE_2D, input_channels=256)
    head = AnchorHeadSingle(m.DENSE_HEAD, input_channels=bev.num_bev_features, num_class=1, class_names=["Car"], grid_size=np.array([1408, 1600, 40]),
                            point_cloud_range=np.array(cfg.DATA_CONFIG.POINT_CLOUD_RANGE, dtype=np.float32), predict_boxes_when_training=True)
    common.init_by_name(bev)
    common.init_by_name(head)
    bev.train()
    head.train()
    d = {"spatial_features": torch.from_numpy(sf), "gt_boxes": torch.from_numpy(gt), "batch_size": 2}
    with torch.no_grad():
        d = head(bev(d))
        loss, tb = head.get_loss()
    f = head.forward_ret_dict
    common.put_digest(gold, "spatial_features_2d", d["spatial_features_2d"].numpy(), n=40000)
    for k in ("cls_preds", "box_preds", "dir_cls_preds"):
        common.put_digest(gold, k, f[k].numpy(), n=40000)
    gold["box_cls_labels"] = f["box_cls_labels"].numpy().astype(np.int8)
    rt = f["box_reg_targets"].numpy()
    nz = np.nonzero(np.abs(rt).sum(-1))
    gold["reg_rows"] = np.stack(nz, 1).astype(np.int32)
    gold["reg_vals"] = rt[nz]
    gold["reg_weights_sum"] = np.array(f["reg_weights"].sum(1).numpy())
    gold["loss"] = np.array([float(loss), tb["rpn_loss_cls"], tb["rpn_loss_loc"], tb["rpn_loss_dir"]], dtype=np.float64)
    common.put_digest(gold, "batch_box_preds", d["batch_box_preds"].numpy(), n=40000)
    common.put_digest(gold, "batch_cls_preds", d["batch_cls_preds"].numpy(), n=40000)
    for mode in ("TRAIN", "TEST"):
        nd = RoIHeadTemplate.proposal_layer(object(), {"batch_size": 2, "batch_box_preds": d["batch_box_preds"].clone(),
                                                       "batch_cls_preds": d["batch_cls_preds"].clone()}, m.ROI_HEAD.NMS_CONFIG[mode])
        gold["rois_" + mode], gold["roi_scores_" + mode], gold["roi_labels_" + mode] = nd["rois"].numpy(), nd["roi_scores"].numpy(), nd["roi_labels"].numpy()
        print(mode, "kept", int((nd["roi_scores"] != 0).sum()), "of", nd["rois"].shape)
    print("loss", gold["loss"], "positives", int((gold["box_cls_labels"] > 0).sum()), "ignored", int((gold["box_cls_labels"] < 0).sum()))
    np.savez_compressed(os.path.join(HERE, "head.npz"), **gold)
    print("wrote head.npz %.0f KB" % (os.path.getsize(os.path.join(HERE, "head.npz")) / 1024))
