"""GPU tests of the module layer (DataProcessor, VFEs, backbones, OccHead3D losses, PassOccVox, the assembled
hot path) against the golden vectors of the real reference and the CPU oracle."""
import numpy as np
import pytest
import torch

import common
from golden_batch import golden_batch
from oracle import occ_oracle, oracle as orc
from test_oracle_golden import canon_slots

from btcdet_amd import synth
from btcdet_amd.config import load_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def G():
    g, scenes, bd = golden_batch()
    cfg = load_cfg()
    from btcdet_amd.btc_path import BtcHotPath
    torch.manual_seed(0)
    model = BtcHotPath(cfg, device=DEV).to(DEV)
    return g, scenes, bd, cfg, model


def voxel_dict(coords, num, voxels):
    return {tuple(c): (int(n), v[:int(n)]) for c, n, v in zip(coords.tolist(), num.tolist(), voxels)}


def test_processor_forward_batch_vs_golden(G):
    g, scenes, bd, cfg, model = G
    proc = model.dataset.data_processor
    pts = np.concatenate([s["points"] for s in scenes])
    pre = np.concatenate([s["pre_rot_points"] for s in scenes])
    offs = np.cumsum([0] + [s["points"].shape[0] for s in scenes]).astype(np.int32)
    rot = np.array([s["rot_z"] for s in scenes], np.float32)
    out = proc.forward_batch(torch.from_numpy(pts).to(DEV), torch.from_numpy(pre).to(DEV), torch.from_numpy(offs).to(DEV),
                             torch.from_numpy(rot).to(DEV))
    # detection grid: no transcendental involved -> bit-exact with the reference's DataProcessor output
    np.testing.assert_array_equal(out["det_voxel_coords"].cpu().numpy(), bd["det_voxel_coords"].numpy().astype(np.int32))
    np.testing.assert_array_equal(out["det_voxel_num_points"].cpu().numpy(), bd["det_voxel_num_points"].numpy().astype(np.int32))
    np.testing.assert_array_equal(out["det_voxels"].cpu().numpy(), bd["det_voxels"].numpy())
    # cylinder grid: device atan2f vs numpy's -> a point on a cell boundary may move; <= 0.2% of the voxels may differ,
    # matched voxels hold the same points within 2e-5 (deg / m)
    a = voxel_dict(out["voxel_coords"].cpu().numpy(), out["voxel_num_points"].cpu().numpy(), out["voxels"].cpu().numpy())
    b = voxel_dict(bd["voxel_coords"].numpy().astype(np.int32), bd["voxel_num_points"].numpy().astype(np.int32), bd["voxels"].numpy())
    bad = len(set(a) ^ set(b))
    for k in set(a) & set(b):
        if a[k][0] != b[k][0] or not np.allclose(a[k][1], b[k][1], rtol=0, atol=2e-5):
            bad += 1
    assert bad <= 0.002 * len(b) + 2, (bad, len(b))


def test_vfe_modules(G):
    g, scenes, bd, cfg, model = G
    d = {"voxels": torch.from_numpy(g["tgt_voxels_absxyz"]).to(DEV), "voxel_num_points": bd["voxel_num_points"].to(DEV)}
    out = model.occ_modules.vfe(d)["voxel_features"]
    np.testing.assert_allclose(out.cpu().numpy(), g["meanvfe_voxel_features"], rtol=1e-6, atol=1e-6)
    d = {"voxels": torch.from_numpy(g["pov_voxels"]).to(DEV), "voxel_num_points": torch.from_numpy(g["pov_voxel_num_points"]).to(DEV),
         "voxel_coords": torch.from_numpy(g["pov_voxel_coords"]).to(DEV)}
    d = model.det_modules.vfe(d)
    np.testing.assert_allclose(d["voxel_features"].cpu().numpy(), g["occvfe_voxel_features"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(d["occ_voxel_features"].cpu().numpy(), g["occvfe_occ_voxel_features"])


@pytest.mark.parametrize("num_dtype", [torch.float32, torch.int32, torch.int64])
def test_vfe_kernels_equal_torch_formulation(num_dtype):
    """csrc/vfe.hip (one launch per encoder) against the module's torch formulation (the restatement pinned by the goldens)"""
    from btcdet_amd import vfe
    from btcdet_amd.config import load_cfg
    cfg = load_cfg()
    torch.manual_seed(3)
    M, P = 5000, 5
    num = torch.randint(0, P + 1, (M,), device=DEV)
    vox = torch.randn(M, P, 6, device=DEV)
    vox[..., 4] = torch.rand(M, P, device=DEV) * (torch.rand(M, P, device=DEV) > 0.6)   # code channels: 0 for raw points
    vox[..., 5] = (vox[..., 4] > 0).float()
    vox = vox * (torch.arange(P, device=DEV).view(1, -1) < num.view(-1, 1)).unsqueeze(-1)  # padding slots are zero
    occ = vfe.OccVFE(cfg.MODEL.VFE, num_point_features=6, data_cfg=cfg.DATA_CONFIG, maxprob=True)
    mean = vfe.MeanVFE(cfg.MODEL.OCC.VFE, num_point_features=4, data_cfg=cfg.DATA_CONFIG, maxprob=False)
    res = {}
    for fused in (True, False):
        vfe.FUSED = fused
        try:
            d = occ({"voxels": vox, "voxel_num_points": num.to(num_dtype)})
            m = mean({"voxels": vox[..., :4].contiguous(), "voxel_num_points": num.to(num_dtype)})
        finally:
            vfe.FUSED = True
        res[fused] = (d["voxel_features"], d["occ_voxel_features"], m["voxel_features"])
    for a, b in zip(res[True], res[False]):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)


def reference_side_dict(g, bd, cfg, device):
    """batch_dict as the reference has it right after OccTargets3D + the synthetic head outputs (oracle = pinned)"""
    O = occ_oracle.OccOracle(cfg)
    t = O.targets(bd)
    d = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in bd.items()}
    d.update({k: v.to(device) for k, v in t.items() if not k.startswith("_") and torch.is_tensor(v)})
    logit, res = common.synthetic_head_outputs(2, O.nz, O.ny, O.nx)
    d["pred_occ_logit"] = torch.from_numpy(logit).to(device)
    d["batch_pred_occ_prob"] = torch.softmax(d["pred_occ_logit"], dim=1)[:, 1] * d["general_cls_loss_mask"]
    d["pred_sem_residuals"] = torch.from_numpy(res).to(device)
    return d


@pytest.mark.parametrize("fused", [True, False])
def test_occ_loss_vs_golden(G, fused):
    """loss values against the reference's golden numbers; gradients of the fused kernels against autograd through
    the reference's op chain (fp64)"""
    from btcdet_amd import occ_head
    g, scenes, bd, cfg, model = G
    d = reference_side_dict(g, bd, cfg, DEV)
    d["pred_occ_logit"].requires_grad_(True)
    d["pred_sem_residuals"].requires_grad_(True)
    occ_head.FUSED_LOSS = fused
    try:
        loss, tb = model.occ_modules.occ_dense_head.get_loss(d)
    finally:
        occ_head.FUSED_LOSS = True
    np.testing.assert_allclose([float(loss), float(tb["occ_loss_cls"]), float(tb["occ_loss_res"])], g["head_loss"], rtol=2e-5)
    loss.backward()
    d64 = {k: (v.detach().cpu().double() if torch.is_tensor(v) and v.is_floating_point() else (v.cpu() if torch.is_tensor(v) else v)) for k, v in d.items()}
    d64["pred_occ_logit"].requires_grad_(True)
    d64["pred_sem_residuals"].requires_grad_(True)
    ref, _, _ = occ_oracle.occ_losses(d64, cfg.MODEL.OCC.OCC_DENSE_HEAD.LOSS_CONFIG.LOSS_WEIGHTS)
    ref.backward()
    np.testing.assert_allclose(d["pred_occ_logit"].grad.cpu().numpy(), d64["pred_occ_logit"].grad.numpy(), rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(d["pred_sem_residuals"].grad.cpu().numpy(), d64["pred_sem_residuals"].grad.numpy(), rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("fused", [True, False])
def test_pass_occ_vox_vs_golden(G, fused):
    """fused = the btc_pass_occ_vox_* path (radix-select top-k + virtual-point re-voxelization);
    not fused = the torch op chain + btc_revoxelize_*; both against the reference's golden output"""
    from btcdet_amd import pass_occ_vox as pov
    g, scenes, bd, cfg, model = G
    d = reference_side_dict(g, bd, cfg, DEV)
    pov.FUSED = fused
    try:
        d = model.occ_modules.occ_pnt_update(d)
    finally:
        pov.FUSED = True
    vc, vn, vv = d["voxel_coords"].cpu().numpy(), d["voxel_num_points"].cpu().numpy(), d["voxels"].cpu().numpy()
    assert d["voxel_coords"].dtype == torch.int64 and d["voxel_num_points"].dtype == torch.int64
    # added occupancy points: same set of (scene, prob) and xyz within 1e-4 m (device cos/sin); order is unspecified in the
    # reference (topk sorted=False), so compare after sorting by (scene, prob, x)
    def srt(p, b):
        k = np.lexsort((p[:, 0], np.round(p[:, 3], 5), b))
        return p[k], b[k]
    pa, ba = srt(d["occ_pnts"].cpu().numpy(), d["added_occ_b_ind"].cpu().numpy())
    pb, bb = srt(g["pov_occ_pnts"], g["pov_added_occ_b_ind"])
    np.testing.assert_array_equal(ba, bb)
    np.testing.assert_allclose(pa[:, 3], pb[:, 3], rtol=0, atol=3e-7)   # softmax evaluated on the device
    np.testing.assert_allclose(pa[:, :3], pb[:, :3], rtol=0, atol=1e-4)
    # merged detection voxels: identical cells / counts unless an added point sits within 1e-4 m of a 5 cm cell face
    a = voxel_dict(vc, vn, canon_slots(vv, vn))
    b = voxel_dict(g["pov_voxel_coords"], g["pov_voxel_num_points"], canon_slots(g["pov_voxels"], g["pov_voxel_num_points"]))
    bad = len(set(a) ^ set(b))
    for k in set(a) & set(b):
        if a[k][0] != b[k][0] or not np.allclose(a[k][1], b[k][1], rtol=0, atol=1e-4):
            bad += 1
    assert bad <= 0.002 * len(b) + 2, (bad, len(b))
    # lexicographic order of the cells (torch.unique(dim=0, sorted=True) in the reference)
    lin = ((vc[:, 0] * 40 + vc[:, 1]) * 1600 + vc[:, 2]) * 1408 + vc[:, 3]
    assert np.all(np.diff(lin) > 0)


def test_occ_backbone_vs_oracle_chain(G):
    """VoxelBackBoneDeconv (eval-mode BN) against the oracle's rulebooks + conv chain, layer table of
    spconv_backbone.py:106-128; checks rulebook construction, the transposed convs and indice_key handling end to end."""
    g, scenes, bd, cfg, model = G
    bb = model.occ_modules.backbone_3d.eval()
    feat = g["meanvfe_voxel_features"]
    idx = bd["voxel_coords"].numpy().astype(np.int32)
    d = {"voxel_features": torch.from_numpy(feat).to(DEV), "voxel_coords": torch.from_numpy(idx).to(DEV), "batch_size": 2}
    with torch.no_grad():
        out = bb(d)["encoded_spconv_tensor"]
    layers = [("conv1.0.0", 1, orc.MODE_CONV), ("conv2.0.0", 2, orc.MODE_CONV), ("conv2.1.0", 1, orc.MODE_SUBM),
              ("conv3.0.0", 2, orc.MODE_CONV), ("conv3.1.0", 1, orc.MODE_SUBM), ("deconv4.0.0", 2, orc.MODE_TRANSPOSE),
              ("deconv4.1.0", 1, orc.MODE_SUBM), ("deconv5.0.0", 2, orc.MODE_TRANSPOSE), ("deconv5.1.0", 1, orc.MODE_SUBM)]
    sd = {k: v.cpu().numpy() for k, v in bb.state_dict().items()}
    shape, f = [9, 157, 209], feat
    bn = np.float32(1.0) / np.sqrt(np.float32(1.0) + np.float32(1e-3))
    for name, stride, mode in layers:
        o_idx, nbr_out, nbr_in, osh = orc.rulebook(idx, shape, 3, stride, 0 if mode == orc.MODE_SUBM else 1, 1, mode)
        f = np.maximum(orc.conv_fwd(f, sd[name + ".weight"], None, nbr_out) * bn, 0).astype(np.float32)
        idx, shape = o_idx, list(osh)
    assert list(out.spatial_shape) == [9, 157, 209] == shape
    np.testing.assert_array_equal(out.indices.cpu().numpy(), idx)
    np.testing.assert_allclose(out.features.cpu().numpy(), f, rtol=1e-4, atol=1e-5)
    bb.train()


def test_full_step_runs_and_is_finite(G):
    import bench
    g, scenes, bd, cfg, model = G
    model.train()
    opts = [torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)]
    batch = bench.build_batches(1, 0, torch.device(DEV))[0]
    step = bench.make_step(model, model, model.dataset.data_processor, opts)
    losses = [float(step(batch)) for _ in range(3)]
    assert all(np.isfinite(losses)), losses
    missing = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
    assert not missing, missing
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_spconv_drop_in_import():
    """`import spconv` resolves to this implementation after install_as_spconv() (spconv_backbone.py:3)"""
    import btcdet_amd
    sp = btcdet_amd.install_as_spconv()
    import spconv
    from spconv.utils import VoxelGeneratorV2
    assert spconv is sp and VoxelGeneratorV2 is sp.utils.VoxelGeneratorV2
    x = spconv.SparseConvTensor(torch.randn(5, 4, device=DEV), torch.tensor([[0, 1, 2, 3], [0, 1, 2, 4], [0, 2, 2, 3], [0, 0, 0, 0], [0, 3, 5, 7]],
                                dtype=torch.int32, device=DEV), [4, 6, 8], 1)
    net = spconv.SparseSequential(spconv.SubMConv3d(4, 8, 3, padding=1, bias=False, indice_key="a"), torch.nn.BatchNorm1d(8), torch.nn.ReLU(),
                                  spconv.SparseConv3d(8, 8, 3, stride=2, padding=1, indice_key="b"),
                                  spconv.SparseInverseConv3d(8, 4, 3, indice_key="b")).to(DEV)
    y = net(x)
    assert y.features.shape == (5, 4) and torch.equal(y.indices, x.indices) and list(y.dense().shape) == [1, 4, 4, 6, 8]


def test_merged_occ_heads_equal_separate_heads(G):
    """OccHead3D with conv_cls / conv_res run as one sparse conv == the two separate convs (forward bit-exact: each output
    column is the same fmaf chain; parameter gradients within fp32 tolerance)"""
    from btcdet_amd import occ_head, spconv
    g, scenes, bd, cfg, model = G
    head = model.occ_modules.occ_dense_head
    torch.manual_seed(3)
    idx = torch.unique(torch.stack([torch.randint(0, 2, (6000,)), torch.randint(0, 9, (6000,)), torch.randint(0, 157, (6000,)),
                                    torch.randint(0, 209, (6000,))], 1), dim=0).int().to(DEV)
    feat = torch.randn(idx.shape[0], 32, device=DEV)
    mask = torch.ones(2, 9, 157, 209, dtype=torch.uint8, device=DEV)
    outs = {}
    for merged in (True, False):
        occ_head.MERGE_HEADS = merged
        head.zero_grad()
        x = spconv.SparseConvTensor(feat.clone().requires_grad_(True), idx, [9, 157, 209], 2)
        d = head({"encoded_spconv_tensor": x, "general_cls_loss_mask": mask})
        (d["pred_occ_logit"].pow(2).sum() + d["pred_sem_residuals"].pow(2).sum()).backward()
        outs[merged] = (d["pred_occ_logit"].detach(), d["pred_sem_residuals"].detach(), x.features.grad.clone(),
                        head.conv_cls[0].weight.grad.clone(), head.conv_res[0].weight.grad.clone(), head.conv_cls[0].bias.grad.clone())
    occ_head.MERGE_HEADS = True
    a, b = outs[True], outs[False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for i in (2, 3, 4, 5):
        np.testing.assert_allclose(a[i].cpu().numpy(), b[i].cpu().numpy(), rtol=2e-4, atol=2e-4)


def test_hot_path_eval_mode():
    """tools/test.py's mode: model.eval(), is_train False -- BatchNorm uses (and does not touch) its running statistics,
    PassOccVox applies EVAL_MAX_NUM_OCC_PNTS but -- like the reference, which computes an eval threshold and then masks with
    self.occ_thresh anyway (add_occ_template.py:100-103) -- OCC_THRESH in both modes, OccTargets3D adds neg_mask
    (occ_targets_template.py:377-379), and two runs of the same batch are identical"""
    import bench
    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    cfg = load_cfg()
    torch.manual_seed(0)
    model = BtcHotPath(cfg, device=torch.device(DEV)).to(DEV)
    batch = bench.build_batches(1, 3, torch.device(DEV))[0]
    proc = model.dataset.data_processor

    def run(train):
        bd = proc.forward_batch(batch["points"], batch["pre_rot_points"], batch["scene_offsets"], batch["rot_z"])
        bd.update({"batch_size": 2, "points": batch["points5"], "gt_boxes": batch["gt_boxes"], "gt_boxes_num": batch["gt_boxes_num"],
                   "box_mirr_flag": batch["box_mirr_flag"], "bm_points": batch["bm_points"], "rot_z": batch["rot_z"], "is_train": train})
        return model(bd)

    model.train()
    for _ in range(2):  # give the running statistics something other than their initial values
        run(True)
    model.eval()
    buffers = {k: v.clone() for k, v in model.named_buffers()}
    with torch.no_grad():
        ret1, _, out1 = run(False)
        ret2, _, out2 = run(False)
    for k, v in model.named_buffers():
        assert torch.equal(v, buffers[k]), k                                    # eval never updates running stats / counters
    assert torch.equal(ret1["spatial_features"], ret2["spatial_features"])      # deterministic
    assert torch.isfinite(ret1["spatial_features"]).all()
    assert "neg_mask" in out1
    occ_pnts = out1["occ_pnts"]
    assert occ_pnts.shape[0] <= 2 * cfg.MODEL.OCC.PARAMS.EVAL_MAX_NUM_OCC_PNTS
    if occ_pnts.shape[0] > 1:
        assert float(occ_pnts[:, 3].min()) > cfg.MODEL.OCC.PARAMS.OCC_THRESH      # the reference's quirk: not EVAL_OCC_THRESH
    # train-mode thresholds admit more cells than eval-mode ones on the same probabilities
    model.train()
    with torch.no_grad():
        _, _, out_t = run(True)
    assert out_t["occ_pnts"].shape[0] <= 2 * cfg.MODEL.OCC.PARAMS.MAX_NUM_OCC_PNTS


@pytest.mark.parametrize("case", ["ties_capped", "all_kept", "one_scene_off"])
def test_pass_occ_vox_multi_workgroup_topk_equals_single_workgroup(G, case):
    """the top-k of PassOccVox spread over (chunks x scenes) workgroups (pov_hist / pov_chunk_*) against the one-workgroup
    pov_select (BTC_TUNE_POV_SELECT = 1): same cells, same order, with heavy ties at the cut (probabilities quantised to
    1/64), with fewer candidates than the cap, and with a scene switched off"""
    from btcdet_amd._lib import lib, check
    g, scenes, bd, cfg, model = G
    mod = model.occ_modules.occ_pnt_update
    gen = torch.Generator(device="cpu").manual_seed(11)
    outs = []
    for single in (1, 0):
        d = reference_side_dict(g, bd, cfg, DEV)
        shape = d["batch_pred_occ_prob"].shape
        p = torch.rand(shape, generator=torch.Generator(device="cpu").manual_seed(5))
        if case == "all_kept":
            p = torch.where(torch.rand(shape, generator=gen) < 1e-3, 0.5 + 0.5 * p, 0.25 * p)   # a few hundred candidates per scene
        else:
            p = torch.round(p * 64) / 64                                                         # ~ncell/128 cells per distinct value
        gen.manual_seed(11)
        d["batch_pred_occ_prob"] = p.to(DEV)
        if case == "one_scene_off":
            d["use_occ_prob"] = np.array([True, False])
        check(lib().btc_tune_set(7, single), "btc_tune_set")
        try:
            d = mod(d)
        finally:
            check(lib().btc_tune_set(7, 0), "btc_tune_set")
        outs.append({k: d[k].cpu().numpy() for k in ("occ_pnts", "added_occ_b_ind", "voxels", "voxel_coords", "voxel_num_points")})
    a, b = outs
    n_sel = a["occ_pnts"].shape[0]
    cap = mod.max_add_occpnts_num
    if case == "ties_capped":
        assert n_sel == 2 * cap
    elif case == "all_kept":
        assert 0 < n_sel < cap
    else:
        assert n_sel == cap and set(a["added_occ_b_ind"].tolist()) == {0}
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
