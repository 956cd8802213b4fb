"""BEV backbone + anchor head + proposal step on the GPU against the reference's OWN modules run on CPU
(tests/golden/gen_head_golden.py -> head.npz): same name-keyed weights, same hash-generated BEV map, same boxes.
Dense convs run on the vendor library on both sides (different algorithms: fp32 tolerance); target assignment is exact;
proposals go through this repository's rotated-NMS kernel."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import common  # noqa: E402

from btcdet_amd.config import load_cfg  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class ED(dict):
    __getattr__ = dict.get
    __setattr__ = dict.__setitem__

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = ED(v) if isinstance(v, dict) else ([ED(x) if isinstance(x, dict) else x for x in v] if isinstance(v, list) else v)


BEV_CFG = ED(LAYER_NUMS=[5, 5], LAYER_STRIDES=[1, 2], NUM_FILTERS=[128, 128], UPSAMPLE_STRIDES=[1, 2], NUM_UPSAMPLE_FILTERS=[128, 128])
HEAD_CFG = ED(CLASS_AGNOSTIC=False, USE_DIRECTION_CLASSIFIER=True, DIR_OFFSET=0.78539, DIR_LIMIT_OFFSET=0.0, NUM_DIR_BINS=2,
              ANCHOR_GENERATOR_CONFIG=[dict(class_name="Car", anchor_sizes=[[3.9, 1.6, 1.56]], anchor_rotations=[0, 1.57], anchor_bottom_heights=[-1.78],
                                            align_center=False, feature_map_stride=8, matched_threshold=0.6, unmatched_threshold=0.45)],
              TARGET_ASSIGNER_CONFIG=dict(NAME="AxisAlignedTargetAssigner", POS_FRACTION=-1.0, SAMPLE_SIZE=512, NORM_BY_NUM_EXAMPLES=False,
                                          MATCH_HEIGHT=False, BOX_CODER="ResidualCoder"),
              LOSS_CONFIG=dict(LOSS_WEIGHTS=dict(cls_weight=1.0, loc_weight=2.0, dir_weight=0.2, code_weights=[1.0] * 7)))
NMS = {"TRAIN": ED(NMS_TYPE="nms_gpu", MULTI_CLASSES_NMS=False, NMS_PRE_MAXSIZE=9000, NMS_POST_MAXSIZE=256, NMS_THRESH=0.8),
       "TEST": ED(NMS_TYPE="nms_gpu", MULTI_CLASSES_NMS=False, NMS_PRE_MAXSIZE=1024, NMS_POST_MAXSIZE=100, NMS_THRESH=0.7)}


def test_bev_backbone_anchor_head_and_proposals_vs_reference():
    from btcdet_amd.bev_backbone import BaseBEVBackbone
    from btcdet_amd.dense_head import AnchorHeadSingle, proposal_layer
    g = np.load(os.path.join(HERE, "golden", "head.npz"))
    sf = (common._hash01(2 * 256 * 200 * 176, 9) - np.float32(0.35)).clip(0).reshape(2, 256, 200, 176)
    bev = BaseBEVBackbone(BEV_CFG, input_channels=256)
    head = AnchorHeadSingle(HEAD_CFG, input_channels=bev.num_bev_features, num_class=1, class_names=["Car"], grid_size=np.array([1408, 1600, 40]),
                            point_cloud_range=np.array([0, -40, -3, 70.4, 40, 1], dtype=np.float32))
    common.init_by_name(bev)
    common.init_by_name(head)
    bev, head = bev.to(DEV).train(), head.to(DEV).train()
    d = {"spatial_features": torch.from_numpy(sf).to(DEV), "gt_boxes": torch.from_numpy(g["gt_boxes"]).to(DEV), "batch_size": 2}
    with torch.no_grad():
        d = head(bev(d))
        loss, tb = head.get_loss()
    f = head.forward_ret_dict
    for key, t in (("spatial_features_2d", d["spatial_features_2d"]), ("cls_preds", f["cls_preds"]), ("box_preds", f["box_preds"]),
                   ("dir_cls_preds", f["dir_cls_preds"]), ("batch_cls_preds", d["batch_cls_preds"]), ("batch_box_preds", d["batch_box_preds"])):
        scale = float(np.abs(g[key + "__sample"]).max())
        err, _ = common.check_digest(g, key, t.float().cpu().numpy(), rtol=0, atol=3e-4 * scale, what=key, sum_rtol=1e-4)
        print("%s: max |diff| %.2e of scale %.2e" % (key, err, scale))
    # target assignment: labels exactly, regression targets to fp32 rounding
    np.testing.assert_array_equal(f["box_cls_labels"].cpu().numpy().astype(np.int8), g["box_cls_labels"])
    rt = f["box_reg_targets"].cpu().numpy()
    rows = np.stack(np.nonzero(np.abs(rt).sum(-1)), 1).astype(np.int32)
    np.testing.assert_array_equal(rows, g["reg_rows"])
    np.testing.assert_allclose(rt[rows[:, 0], rows[:, 1]], g["reg_vals"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(f["reg_weights"].sum(1).cpu().numpy(), g["reg_weights_sum"], rtol=0, atol=0)
    got = [float(loss), tb["rpn_loss_cls"], tb["rpn_loss_loc"], tb["rpn_loss_dir"]]
    print("rpn losses", got, "reference", list(g["loss"]))
    np.testing.assert_allclose(got, g["loss"], rtol=2e-3)
    # proposals: the HIP NMS on the device's own boxes vs the reference's on its boxes -- same set up to fp32 noise in the scores
    for mode in ("TRAIN", "TEST"):
        nd = proposal_layer({"batch_size": 2, "batch_box_preds": d["batch_box_preds"].clone(), "batch_cls_preds": d["batch_cls_preds"].clone()}, NMS[mode])
        rois, ref = nd["rois"].cpu().numpy(), g["rois_" + mode]
        assert rois.shape == ref.shape and nd["roi_labels"].cpu().numpy().shape == g["roi_labels_" + mode].shape
        miss = 0
        for b in range(2):
            a, r = rois[b], ref[b]
            dist = np.abs(a[:, None, :] - r[None, :, :]).max(-1)
            miss += int((dist.min(1) > 2e-3).sum())
        print("%s proposals without a reference proposal within 2e-3: %d of %d" % (mode, miss, rois.shape[0] * rois.shape[1]))
        assert miss <= 0.03 * rois.shape[0] * rois.shape[1]
        np.testing.assert_allclose(np.sort(nd["roi_scores"].cpu().numpy(), 1)[:, -20:], np.sort(g["roi_scores_" + mode], 1)[:, -20:], rtol=1e-3, atol=1e-3)
        assert set(np.unique(nd["roi_labels"].cpu().numpy())) <= {1}
