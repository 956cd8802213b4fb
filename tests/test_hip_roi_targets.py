"""ROI-head training targets and losses (btcdet_amd/roi_targets.py, ConvHead.assign_targets / get_loss) against the reference's own
RoIHeadTemplate / ProposalTargetLayer run on CPU (tests/golden/gen_roi_targets_golden.py -> roi_targets.npz): three scenes of 512
proposals (foreground / hard / easy background thirds; the third scene has no boxes), the reference's own draw handed over, so that
everything deterministic is compared -- and the device-side sampler's quotas and candidate sets on its own draws."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import common  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _gold():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "roi_targets.npz"))


def _inputs():
    inp = common.roi_target_inputs()
    return {k: (torch.from_numpy(v).to(DEV) if isinstance(v, np.ndarray) else v) for k, v in inp.items()}


def _cfg():
    from btcdet_amd.config import load_cfg
    return load_cfg().MODEL.ROI_HEAD


def test_matching_targets_and_canonical_transform_vs_reference():
    from btcdet_amd.roi_targets import ProposalTargetLayer, canonical_targets, match_rois
    g, bd, cfg = _gold(), _inputs(), _cfg()
    for b in range(bd["batch_size"]):
        ov, _ = match_rois(bd["rois"][b], bd["roi_labels"][b], bd["gt_boxes"][b])
        np.testing.assert_allclose(ov.cpu().numpy(), g["max_overlaps"][b], rtol=0, atol=2e-5)      # rotated-overlap kernel vs the oracle's clipper
    sel = torch.from_numpy(g["sampled_inds"]).to(DEV)
    t = canonical_targets(ProposalTargetLayer(cfg.TARGET_CONFIG)(bd, sampled_inds=sel))
    for k in ("rois", "roi_scores", "gt_of_rois_src"):
        assert np.array_equal(t[k].cpu().numpy(), g["t_" + k]), k
    assert np.array_equal(t["roi_labels"].cpu().numpy(), g["t_roi_labels"])
    np.testing.assert_allclose(t["gt_iou_of_rois"].cpu().numpy(), g["t_gt_iou_of_rois"], rtol=0, atol=2e-5)
    # labels / masks: equal wherever the IoU is not within the kernel's 2e-5 of a threshold
    iou = g["t_gt_iou_of_rois"]
    clear = np.ones_like(iou, bool)
    for th in (cfg.TARGET_CONFIG.REG_FG_THRESH, cfg.TARGET_CONFIG.CLS_FG_THRESH, cfg.TARGET_CONFIG.CLS_BG_THRESH):
        clear &= np.abs(iou - th) > 1e-4
    assert clear.mean() > 0.99
    assert np.array_equal(t["reg_valid_mask"].cpu().numpy()[clear], g["t_reg_valid_mask"][clear])
    np.testing.assert_allclose(t["rcnn_cls_labels"].cpu().numpy()[clear], g["t_rcnn_cls_labels"][clear], rtol=0, atol=1e-4)
    np.testing.assert_allclose(t["gt_of_rois"].cpu().numpy(), g["t_gt_of_rois"], rtol=0, atol=2e-5)


def test_losses_vs_reference():
    from btcdet_amd.dense_head import ResidualCoder
    from btcdet_amd.roi_targets import rcnn_cls_loss, rcnn_reg_loss
    g, cfg = _gold(), _cfg()
    B, R = g["t_rois"].shape[:2]
    ret = {k[2:]: torch.from_numpy(g[k]).to(DEV) for k in g.files if k.startswith("t_")}
    cls = (common._hash01(B * R, 320) - np.float32(0.5)) * np.float32(4.0)
    reg = (common._hash01(B * R * 7, 321).reshape(B * R, 7) - np.float32(0.5)) * np.float32(0.6)
    ret["rcnn_cls"] = torch.from_numpy(cls.reshape(B * R, 1)).to(DEV).requires_grad_(True)
    ret["rcnn_reg"] = torch.from_numpy(reg).to(DEV).requires_grad_(True)
    l_cls = rcnn_cls_loss(ret["rcnn_cls"], ret["rcnn_cls_labels"], cfg.LOSS_CONFIG)
    l_reg, corner = rcnn_reg_loss(ret, ResidualCoder(), cfg.LOSS_CONFIG)
    total, ref_cls, ref_reg, ref_corner = g["loss"]
    np.testing.assert_allclose([float(l_cls), float(l_reg - corner), float(corner), float(l_cls + l_reg)], [ref_cls, ref_reg, ref_corner, total], rtol=2e-5)
    (l_cls + l_reg).backward()
    assert torch.isfinite(ret["rcnn_cls"].grad).all() and torch.isfinite(ret["rcnn_reg"].grad).all() and float(ret["rcnn_reg"].grad.abs().sum()) > 0


def test_device_sampler_quotas_and_sets():
    """proposal_target_layer.py:117-197 on the device's own draws: foreground slots first, without replacement, min(quota, candidates);
    background slots hard : easy in the configured ratio (hard capped by its candidates), every index from the right set; the scene
    without boxes is all easy background"""
    from btcdet_amd.roi_targets import sample_rois
    g, cfg = _gold(), _cfg().TARGET_CONFIG
    R, quota = cfg.ROI_PER_IMAGE, int(round(cfg.FG_RATIO * cfg.ROI_PER_IMAGE))
    gen = torch.Generator(device=DEV)
    gen.manual_seed(3)
    for b in range(g["max_overlaps"].shape[0]):
        ov = torch.from_numpy(g["max_overlaps"][b]).to(DEV)
        fg = (ov >= min(cfg.REG_FG_THRESH, cfg.CLS_FG_THRESH)).cpu().numpy()
        easy = (ov < cfg.CLS_BG_THRESH_LO).cpu().numpy()
        hard = ~fg & ~easy
        seen = set()
        for _ in range(3):
            sel = sample_rois(ov, cfg, gen).cpu().numpy()
            seen.add(tuple(sel))
            n_fg = min(quota, int(fg.sum())) if (easy | hard).any() else R
            assert fg[sel[:n_fg]].all() and not fg[sel[n_fg:]].any()
            if (easy | hard).any():
                assert len(set(sel[:n_fg])) == n_fg                                     # without replacement
            m = R - n_fg
            n_hard = min(int(m * cfg.HARD_BG_RATIO), int(hard.sum())) if (hard.any() and easy.any()) else (m if hard.any() else 0)
            assert hard[sel[n_fg:n_fg + n_hard]].all() and easy[sel[n_fg + n_hard:]].all()
        assert len(seen) == 3                                                            # fresh draws
    # degenerate case of the reference: foreground only -> every slot drawn from it with replacement
    ov = torch.full((40,), 0.9, device=DEV)
    sel = sample_rois(ov, cfg, gen).cpu().numpy()
    assert sel.shape == (R,) and sel.min() >= 0 and sel.max() < 40 and len(set(sel)) > 20


def test_full_heads_training_step_runs():
    """BtcHotPath(heads="full"): RPN -> proposals -> ROI targets -> ConvHead -> rcnn loss, two optimizer steps through HotPathTrainer:
    finite losses, every ROI-head parameter receives a gradient"""
    import bench
    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    from btcdet_amd.trainer import HotPathTrainer
    torch.manual_seed(0)
    model = BtcHotPath(load_cfg(), device=DEV, heads="full").to(DEV).train()
    batches = bench.build_batches(2, 0, DEV)
    out, tb, bd = model(model.prepare(batches[0]))
    loss = model.det_loss(out, bd) + out["loss_occ"]
    loss.backward()
    assert torch.isfinite(loss)
    missing = [n for n, p in model.det_modules.roi_head.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    assert not missing, missing
    model.zero_grad(set_to_none=True)
    tr = HotPathTrainer(model, det_loss=model.det_loss)
    for i in range(3):
        l = tr.step(batches[i % 2], batches[(i + 1) % 2])
    tr.finish()
    assert torch.isfinite(l)
