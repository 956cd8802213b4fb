"""GPU path against the full-size golden vectors of the REAL reference (tests/golden/gen_golden_full.py; the CPU oracle is
pinned to the same vectors by tests/test_golden_full_cpu.py):

  a1 / a2   raw resident scans -> HIP range mask + stable compaction + shuffle: bit-exact vs the reference's masked and
            shuffled points (SHA-1), for full-size, 40-point and empty scenes
  a4 / a5   both voxelizers on those points: detection grid bit-exact incl. the 16 000-voxel cap (coords, counts, payload
            SHA-1); occupancy grid up to device-atan2 boundary cases (measured and printed, bound 2x the observed)
  a6 / a7   collate_device == reference collate_batch + load_data_to_gpu
  a17-a19   occupancy backbone + head, name-keyed weights, train-mode BatchNorm vs the reference's OWN modules run over the
            oracle-backed spconv; loss vs the reference's get_loss on those outputs
  a21 / a22 PassOccVox with the top-k really selecting (> 2048 candidates), OccVFE
  a23 / a24 detection backbone + HeightCompression vs the reference's OWN VoxelBackBone8xOcc.forward (train + eval BN)
"""
import numpy as np
import pytest
import torch

import common
from golden_batch import FULL_TAGS, golden_batch_full
from oracle import occ_oracle

from btcdet_amd.config import load_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
_CACHE = {}


def case(tag):
    if tag not in _CACHE:
        _CACHE[tag] = golden_batch_full(tag)
    return _CACHE[tag]


@pytest.fixture(scope="module")
def model():
    from btcdet_amd.btc_path import BtcHotPath
    cfg = load_cfg()
    torch.manual_seed(0)
    m = BtcHotPath(cfg, device=DEV).to(DEV)
    for mod in (m.occ_modules.backbone_3d, m.occ_modules.occ_dense_head, m.det_modules.backbone_3d):
        common.init_by_name(mod)
    return cfg, m


def voxel_dict(coords, num, voxels):
    return {tuple(c): (int(n), v[:int(n)]) for c, n, v in zip(coords.tolist(), num.tolist(), voxels)}


@pytest.mark.parametrize("tag", FULL_TAGS)
def test_resident_mask_shuffle_voxelize(model, tag):
    cfg, m = model
    g, scenes, bd, host = case(tag)
    proc = m.dataset.data_processor
    raw = np.concatenate([s["raw_points"] for s in scenes])
    raw_pre = np.concatenate([s["raw_pre_rot_points"] for s in scenes])
    offs = np.cumsum([0] + [s["raw_points"].shape[0] for s in scenes]).astype(np.int32)
    rot = np.array([s["rot_z"] for s in scenes], np.float32)
    perms = [g["proc%d_shuffle_idx" % i] for i in range(len(scenes))]
    out = proc.forward_raw_batch(torch.from_numpy(raw).to(DEV), torch.from_numpy(raw_pre).to(DEV), torch.from_numpy(offs).to(DEV),
                                 torch.from_numpy(rot).to(DEV), shuffle_idx=perms)
    # a1 / a2: the masked + shuffled points of every scene, bit for bit
    bounds = out["scene_offsets"].cpu().numpy()
    assert out["scene_counts"] == [s["points"].shape[0] for s in scenes]
    pts, pre = out["masked_points"].cpu().numpy(), out["masked_pre_rot_points"].cpu().numpy()
    for i, s in enumerate(scenes):
        assert np.array_equal(common.sha1(pts[bounds[i]:bounds[i + 1]]), g["proc%d_points_sha1" % i])
        np.testing.assert_array_equal(pre[bounds[i]:bounds[i + 1]], s["masked_pre_rot_points"])      # masked, NOT shuffled
    # a5: detection grid bit-exact (no transcendental involved), cap included
    dc, dn, dv = out["det_voxel_coords"].cpu().numpy(), out["det_voxel_num_points"].cpu().numpy(), out["det_voxels"].cpu().numpy()
    np.testing.assert_array_equal(dc, host["det_voxel_coords"].astype(np.int32))
    np.testing.assert_array_equal(dn, host["det_voxel_num_points"].astype(np.int32))
    lo = 0
    for i, s in enumerate(scenes):
        k = s["det_voxel_coords"].shape[0]
        assert np.array_equal(common.sha1(dv[lo:lo + k]), g["proc%d_det_voxels_sha1" % i])
        lo += k
    # a4: cylinder grid -- device atan2f vs numpy's: a point within an ulp of a cell face may change cell
    a = voxel_dict(out["voxel_coords"].cpu().numpy(), out["voxel_num_points"].cpu().numpy(), out["voxels"].cpu().numpy())
    b = voxel_dict(host["voxel_coords"].astype(np.int32), host["voxel_num_points"].astype(np.int32), host["voxels"])
    bad = len(set(a) ^ set(b))
    for k in set(a) & set(b):
        if a[k][0] != b[k][0] or not np.allclose(a[k][1], b[k][1], rtol=0, atol=2e-5):
            bad += 1
    print("occupancy-grid voxels differing from the reference [%s]: %d of %d" % (tag, bad, len(b)))
    assert bad <= 4, (bad, len(b))            # observed: 0-2 per batch of ~10 k voxels


@pytest.mark.parametrize("tag", FULL_TAGS)
def test_collate_device_vs_reference(tag):
    from btcdet_amd.collate import collate_device
    g, scenes, bd, host = case(tag)
    keys = ["points", "voxels", "voxel_coords", "voxel_num_points", "det_voxels", "det_voxel_coords", "det_voxel_num_points", "gt_boxes",
            "box_mirr_flag", "rot_z"] + (["bm_points"] if bool(g["meta_with_bm_key"]) else [])
    dev_scenes = [{k: torch.from_numpy(np.asarray(s[k])).to(DEV) for k in keys} for s in scenes]
    out = collate_device(dev_scenes)
    for k in keys + ["batch_voxel_num", "batch_det_voxel_num"]:
        assert out[k].dtype == torch.float32, k
        np.testing.assert_array_equal(out[k].cpu().numpy(), np.asarray(host[k], dtype=np.float32), err_msg=k)
    assert out["gt_boxes_num"] == list(host["gt_boxes_num"]) and out["batch_size"] == host["batch_size"]


def test_occ_net_vs_reference_modules(model):
    """VoxelBackBoneDeconv + OccHead3D (+ get_loss) on the reference's MeanVFE output of the full-size batch"""
    cfg, m = model
    g, scenes, bd, host = case("full_a")
    O = occ_oracle.OccOracle(cfg)
    t = O.targets(bd)
    d = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in bd.items()}
    d.update({k: v.to(DEV) for k, v in t.items() if not k.startswith("_") and torch.is_tensor(v)})
    d["voxel_features"] = torch.from_numpy(g["meanvfe_voxel_features"]).to(DEV)
    bb, head = m.occ_modules.backbone_3d, m.occ_modules.occ_dense_head
    bb.train()
    head.train()
    saved = {k: v.clone() for k, v in bb.state_dict().items()}
    with torch.no_grad():
        d = head(bb(d))
        loss, tb = head.get_loss(d)
    bb.load_state_dict(saved)
    x = d["encoded_spconv_tensor"]
    assert x.features.shape[0] == int(g["net_occ_out_n"])
    assert np.array_equal(common.sha1(x.indices.cpu().numpy().astype(np.int32)), g["net_occ_out_indices_sha1"])
    for key, a in (("net_occ_features", x.features), ("net_pred_occ_logit", d["pred_occ_logit"]),
                   ("net_pred_sem_residuals", d["pred_sem_residuals"]), ("net_batch_pred_occ_prob", d["batch_pred_occ_prob"])):
        scale = float(np.abs(g[key + "__sample"]).max())
        err, _ = common.check_digest(g, key, a.float().cpu().numpy(), rtol=0, atol=2e-5 * scale, what=key)
        print("occupancy net vs reference modules, %s: max |diff| %.2e of scale %.2e" % (key, err, scale))
    got = [float(loss), float(tb["occ_loss_cls"]), float(tb["occ_loss_res"])]
    print("occupancy loss on real head outputs: %s vs reference %s" % (got, list(g["net_head_loss"])))
    np.testing.assert_allclose(got, g["net_head_loss"], rtol=5e-5)
    n_cand = [int((d["batch_pred_occ_prob"][b] > cfg.MODEL.OCC.PARAMS.OCC_THRESH).sum()) for b in range(2)]
    assert all(abs(a - int(b)) <= 2 for a, b in zip(n_cand, g["net_n_candidates"])), (n_cand, g["net_n_candidates"])


@pytest.mark.parametrize("tag", FULL_TAGS)
def test_pass_occ_vox_and_occ_vfe_vs_reference(model, tag):
    cfg, m = model
    g, scenes, bd, host = case(tag)
    B = bd["batch_size"]
    O = occ_oracle.OccOracle(cfg)
    t = O.targets(bd)
    d = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in bd.items()}
    d.update({k: v.to(DEV) for k, v in t.items() if not k.startswith("_") and torch.is_tensor(v)})
    logit, res = common.synthetic_head_outputs(B, O.nz, O.ny, O.nx)
    d["pred_occ_logit"] = torch.from_numpy(logit).to(DEV)
    d["batch_pred_occ_prob"] = torch.softmax(d["pred_occ_logit"], dim=1)[:, 1] * d["general_cls_loss_mask"]
    d["pred_sem_residuals"] = torch.from_numpy(res).to(DEV)
    d = m.occ_modules.occ_pnt_update(d)
    vc, vn, vv = d["voxel_coords"].cpu().numpy(), d["voxel_num_points"].cpu().numpy(), d["voxels"].cpu().numpy()

    def srt(p, b):
        k = np.lexsort((p[:, 0], np.round(p[:, 3], 5), b))
        return p[k], b[k]
    pa, ba = srt(d["occ_pnts"].cpu().numpy(), d["added_occ_b_ind"].cpu().numpy())
    pb, bb_ = srt(g["pov_occ_pnts"], g["pov_added_occ_b_ind"])
    np.testing.assert_array_equal(ba, bb_)
    np.testing.assert_allclose(pa[:, 3], pb[:, 3], rtol=0, atol=3e-7)
    np.testing.assert_allclose(pa[:, :3], pb[:, :3], rtol=0, atol=1e-4)
    # merged voxels: cells / counts / per-voxel channel sums; an added point within 1e-4 m of a 5 cm cell face may change cell
    a = {tuple(c): (int(n), s) for c, n, s in zip(vc.tolist(), vn.tolist(), vv.astype(np.float64).sum(1))}
    b = {tuple(c): (int(n), s) for c, n, s in zip(g["pov_voxel_coords"].tolist(), g["pov_voxel_num_points"].tolist(), g["pov_voxel_slot_sums"].astype(np.float64))}
    bad = len(set(a) ^ set(b))
    for k in set(a) & set(b):
        if a[k][0] != b[k][0] or not np.allclose(a[k][1], b[k][1], rtol=1e-5, atol=2e-4):
            bad += 1
    print("merged detection voxels differing from the reference [%s]: %d of %d" % (tag, bad, len(b)))
    assert bad <= 8, (bad, len(b))            # observed 0-3
    lin = ((vc[:, 0] * 40 + vc[:, 1]) * 1600 + vc[:, 2]) * 1408 + vc[:, 3]
    assert np.all(np.diff(lin) > 0)           # lexicographic order (torch.unique(dim=0, sorted=True) in the reference)
    # OccVFE on the reference's merged voxels (exact inputs)
    if bad == 0:
        f = m.det_modules.vfe({"voxels": d["voxels"], "voxel_num_points": d["voxel_num_points"], "voxel_coords": d["voxel_coords"]})
        np.testing.assert_allclose(f["voxel_features"].cpu().numpy(), g["occvfe_voxel_features"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(f["occ_voxel_features"].cpu().numpy(), g["occvfe_occ_voxel_features"], rtol=0, atol=3e-7)


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_det_backbone_vs_reference_module(model, mode):
    cfg, m = model
    g = common.load("full_a")
    bb, bev = m.det_modules.backbone_3d, m.det_modules.map_to_bev_module
    bb.train(mode == "train")
    saved = {k: v.clone() for k, v in bb.state_dict().items()}
    d = {"voxel_features": torch.from_numpy(g["occvfe_voxel_features"]).to(DEV), "occ_voxel_features": torch.from_numpy(g["occvfe_occ_voxel_features"]).to(DEV),
         "voxel_coords": torch.from_numpy(g["pov_voxel_coords"]).to(DEV), "batch_size": 2}
    with torch.no_grad():
        d = bev(bb(d))
    bb.load_state_dict(saved)
    bb.train()
    out, xc = d["encoded_spconv_tensor"], d["multi_scale_3d_features"]["x_combine"]
    p = "net_det_%s_" % mode
    assert np.array_equal(common.sha1(out.indices.cpu().numpy().astype(np.int32)), g[p + "out_indices_sha1"])
    assert np.array_equal(common.sha1(xc.indices.cpu().numpy().astype(np.int32)), g[p + "xc_indices_sha1"])
    for key, a in (("out", out.features), ("x_combine", xc.features), ("spatial_features", d["spatial_features"])):
        scale = float(np.abs(g[p + key + "__sample"]).max())
        err, _ = common.check_digest(g, p + key, a.float().cpu().numpy(), rtol=0, atol=2e-5 * scale, what=key)
        print("detection backbone vs reference module [%s] %s: max |diff| %.2e of scale %.2e" % (mode, key, err, scale))
