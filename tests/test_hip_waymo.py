"""BASELINE.json configs[4]: Waymo-SHAPED synthetic scenes (~166 k points, detection grid 1504 x 1504 x 40, cylinder
occupancy grid 325 x 697 x 9; constants of btcdet_amd/cfgs/btcdet_waymo_synth.yaml -- the reference has no Waymo yaml).
Same bars as the KITTI-sized tests: voxels and rulebooks bit-exact against the C oracle, conv forward / dgrad bit-exact,
wgrad within the stated fp32 tolerance; plus size-independent properties of the full-size rulebooks and one full
forward + backward step of the hot path on this configuration."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WAYMO_CFG = os.path.join(ROOT, "btcdet_amd", "cfgs", "btcdet_waymo_synth.yaml")
OCC_RANGE = [2.24, -180.6624, -2.6, 106.24, 180.6624, 0.64]
OCC_VOXEL = [0.32, 0.5184, 0.36]


def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def waymo_batch():
    from btcdet_amd import synth
    return synth.make_batch([5000, 5001], profile="waymo")


@pytest.fixture(scope="module")
def det_voxels(waymo_batch):
    """oracle voxelization of both scenes on the detection grid -> indices [b,z,y,x]"""
    from btcdet_amd import synth
    og = orc.VoxelGeneratorV2(synth.WAYMO_DET_VOXEL, synth.WAYMO_DET_RANGE, 5, 150000)
    res = [og.generate(s["points"]) for s in waymo_batch["scenes"]]
    idx = np.concatenate([np.pad(r["coordinates"][:r["voxel_num"]], ((0, 0), (1, 0)), constant_values=i)
                          for i, r in enumerate(res)]).astype(np.int32)
    return res, idx


def test_waymo_scene_shape(waymo_batch):
    n = [s["points"].shape[0] for s in waymo_batch["scenes"]]
    assert all(150000 < v < 200000 for v in n), n


def test_waymo_voxelizer_both_grids_bit_exact(waymo_batch, det_voxels):
    from btcdet_amd import synth
    from btcdet_amd.spconv import utils
    scenes = waymo_batch["scenes"]
    for pts_list, vsize, rng_range, max_pts, max_vox in (
            ([s["points"] for s in scenes], synth.WAYMO_DET_VOXEL, synth.WAYMO_DET_RANGE, 5, 150000),
            ([orc.absxyz_2_cylinxyz_np(s["pre_rot_points"]) for s in scenes], OCC_VOXEL, OCC_RANGE, 12, 80000),
            ([s["points"] for s in scenes], synth.WAYMO_DET_VOXEL, synth.WAYMO_DET_RANGE, 2, 50000)):   # both caps bite
        gen = utils.VoxelGeneratorV2(vsize, rng_range, max_pts, max_vox)
        ogen = orc.VoxelGeneratorV2(vsize, rng_range, max_pts, max_vox)
        pts = np.concatenate(pts_list, axis=0).astype(np.float32)
        offs = np.cumsum([0] + [p.shape[0] for p in pts_list]).astype(np.int32)
        v, c, n = gen.generate_batch(torch.from_numpy(pts).to(dev()), torch.from_numpy(offs).to(dev()))
        v, c, n = v.cpu().numpy(), c.cpu().numpy(), n.cpu().numpy()
        row = 0
        for b, p in enumerate(pts_list):
            r = ogen.generate(p)
            m = r["voxel_num"]
            np.testing.assert_array_equal(c[row:row + m, 0], b)
            np.testing.assert_array_equal(c[row:row + m, 1:], r["coordinates"])
            np.testing.assert_array_equal(n[row:row + m], r["num_points_per_voxel"])
            np.testing.assert_array_equal(v[row:row + m], r["voxels"])
            row += m
        assert row == v.shape[0]
    assert list(gen.grid_size) == [1504, 1504, 40]


def test_waymo_rulebooks_bit_exact_and_properties(det_voxels):
    """SubM and stride-2 rulebooks of the first detection-backbone level at full size (~125 k active voxels)"""
    from btcdet_amd.spconv import ops
    _, idx = det_voxels
    shape = [41, 1504, 1504]
    t = torch.from_numpy(idx).to(dev())
    n = idx.shape[0]
    assert n > 100000
    # submanifold 3x3x3
    rb = ops.build_rulebook(t, 2, shape, 3, 1, 1, 1, 0, True, False)
    o_idx, o_out, o_in, _ = orc.rulebook(idx, shape, 3, 1, 0, 1, orc.MODE_SUBM)
    nbr = rb.nbr_out.cpu().numpy()
    np.testing.assert_array_equal(nbr, o_out)
    np.testing.assert_array_equal(rb.nbr_in.cpu().numpy(), o_in)
    # properties: centre offset is the identity; the map is an involution under offset reversal
    np.testing.assert_array_equal(nbr[:, 13], np.arange(n))
    i, k = np.nonzero(nbr >= 0)
    j = nbr[i, k]
    np.testing.assert_array_equal(nbr[j, 26 - k], i)
    # strided conv 3x3x3 s2 p1 and its transposed form
    rb2 = ops.build_rulebook(t, 2, shape, 3, 2, 1, 1, 0, False, False)
    o_idx, o_out, o_in, o_sh = orc.rulebook(idx, shape, 3, 2, 1, 1, orc.MODE_CONV)
    assert list(o_sh) == [21, 752, 752]
    out_idx = rb2.out_indices.cpu().numpy()
    np.testing.assert_array_equal(out_idx, o_idx)
    np.testing.assert_array_equal(rb2.nbr_out.cpu().numpy(), o_out)
    np.testing.assert_array_equal(rb2.nbr_in.cpu().numpy(), o_in)
    # properties: output rows sorted and unique; nbr_in is the transpose of nbr_out; every input row feeds >= 1 output
    key = ((out_idx[:, 0].astype(np.int64) * 21 + out_idx[:, 1]) * 752 + out_idx[:, 2]) * 752 + out_idx[:, 3]
    assert np.all(np.diff(key) > 0)
    no, ni = rb2.nbr_out.cpu().numpy(), rb2.nbr_in.cpu().numpy()
    i, k = np.nonzero(no >= 0)
    np.testing.assert_array_equal(ni[no[i, k], k], i)
    assert (ni >= 0).sum() == (no >= 0).sum() and np.all((ni >= 0).any(1))


@pytest.mark.parametrize("cin,cout", [(6, 16), (16, 32), (64, 64)])
def test_waymo_conv_full_size(det_voxels, cin, cout, exact_conv):
    """forward / dgrad bit-exact and wgrad within 1e-4 of the scale on the full-size stride-2 rulebook"""
    from btcdet_amd.spconv import ops
    _, idx = det_voxels
    shape = [41, 1504, 1504]
    t = torch.from_numpy(idx).to(dev())
    rb = ops.build_rulebook(t, 2, shape, 3, 2, 1, 1, 0, False, False)
    o_idx, o_out, o_in, _ = orc.rulebook(idx, shape, 3, 2, 1, 1, orc.MODE_CONV)
    rng = np.random.default_rng(cin + cout)
    feat = rng.standard_normal((idx.shape[0], cin)).astype(np.float32)
    W = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    dout = rng.standard_normal((o_idx.shape[0], cout)).astype(np.float32)
    f = torch.from_numpy(feat).to(dev()).requires_grad_(True)
    w = torch.from_numpy(W).to(dev()).requires_grad_(True)
    out = ops.indice_conv(f, w, None, rb)
    out.backward(torch.from_numpy(dout).to(dev()))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), orc.conv_fwd(feat, W, None, o_out))
    np.testing.assert_array_equal(f.grad.cpu().numpy(), orc.conv_dgrad(dout, W, o_in))
    ref = orc.conv_wgrad(feat, dout, o_out, W.shape)
    assert np.abs(w.grad.cpu().numpy() - ref).max() <= 1e-4 * (np.abs(ref).max() + 1e-6)


def test_waymo_conv_bf16_operands_full_size(det_voxels):
    """BASELINE configs[4] ("mixed bf16") at full size: bf16 operands on the bf16 matrix pipe on the ~125 k-row stride-2 rulebook,
    within one bf16 ulp + 2e-6 of the scale of the oracle's fp32 chain over the same bf16-rounded operands (forward and dgrad)"""
    from btcdet_amd import _lib
    from btcdet_amd.spconv import ops
    if _lib.fast() is None:
        pytest.skip("compiled binding not built")
    _, idx = det_voxels
    shape = [41, 1504, 1504]
    rb = ops.build_rulebook(torch.from_numpy(idx).to(dev()), 2, shape, 3, 2, 1, 1, 0, False, False)
    o_idx, o_out, o_in, _ = orc.rulebook(idx, shape, 3, 2, 1, 1, orc.MODE_CONV)
    rng = np.random.default_rng(64)
    feat = orc.bf16_round(rng.standard_normal((idx.shape[0], 64)).astype(np.float32))
    W = (rng.standard_normal((3, 3, 3, 64, 64)) / 8.0).astype(np.float32)
    dout = orc.bf16_round(rng.standard_normal((o_idx.shape[0], 64)).astype(np.float32))
    f = torch.from_numpy(feat).to(dev()).to(torch.bfloat16).requires_grad_(True)
    w = torch.from_numpy(W).to(dev()).requires_grad_(True)
    out = ops.indice_conv(f, w, None, rb)
    out.backward(torch.from_numpy(dout).to(dev()).to(torch.bfloat16))
    Wq = orc.bf16_round(W)
    for got, ref in ((out.detach().float().cpu().numpy(), orc.conv_fwd(feat, Wq, None, o_out)),
                     (f.grad.float().cpu().numpy(), orc.conv_dgrad(dout, Wq, o_in))):
        bound = 2.0 ** -8 * np.abs(ref) + 2e-6 * float(np.abs(ref).max())
        assert float((np.abs(got - ref) / bound).max()) <= 1.0


@pytest.mark.parametrize("features", ["fp32", "bf16"])
def test_waymo_hot_path_step(waymo_batch, features):
    """one forward + backward of the whole hot path on the Waymo-shaped configuration (fp32, and configs[4]'s mixed bf16):
    shapes of SURVEY.md §8a scaled to this grid, finite loss, gradients on every parameter"""
    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    b = waymo_batch
    torch.manual_seed(0)
    cfg = load_cfg(WAYMO_CFG)
    if features == "bf16":
        cfg.MODEL.OCC.BACKBONE_3D["FEATURE_DTYPE"] = "bf16"
        cfg.MODEL.BACKBONE_3D["FEATURE_DTYPE"] = "bf16"
    model = BtcHotPath(cfg, device=dev()).to(dev()).train()
    d = dev()
    bd = model.dataset.data_processor.forward_batch(
        torch.from_numpy(np.ascontiguousarray(b["points"][:, 1:])).to(d), torch.from_numpy(b["pre_rot_points"]).to(d),
        torch.from_numpy(b["scene_offsets"]).to(d), torch.from_numpy(b["rot_z"]).to(d))
    bd.update({"batch_size": 2, "points": torch.from_numpy(b["points"]).to(d), "gt_boxes": torch.from_numpy(b["gt_boxes"]).to(d),
               "gt_boxes_num": torch.tensor(b["gt_boxes_num"], dtype=torch.int32, device=d),
               "box_mirr_flag": torch.from_numpy(b["box_mirr_flag"]).to(d), "bm_points": torch.from_numpy(b["bm_points"]).to(d),
               "rot_z": torch.from_numpy(b["rot_z"]).to(d), "is_train": True})
    assert list(model.dataset.det_grid_size) == [1504, 1504, 40] and list(model.dataset.occ_grid_size) == [325, 697, 9]
    ret, tb, out = model(bd)
    loss = ret["loss_occ"] + 1e-3 * ret["spatial_features"].pow(2).mean() + 1e-3 * ret["x_combine"].float().pow(2).mean()
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    assert list(ret["spatial_features"].shape) == [2, 256, 188, 188]
    assert out["batch_pred_occ_prob"].shape == (2, 9, 697, 325)
    missing = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
    assert not missing, missing
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
