"""Parity of the ASSEMBLED detection backbone: VoxelBackBone8xOcc + HeightCompression on the GPU against
the graph restated in tests/det_chain.py from spconv_backbone.py:936-1019 (+ :869-873 sparse_cat, :905-918 res_combine,
:920-933 compress_height) over the oracle's rulebooks / conv / max-pool / dense.

  * indices of every output level: bit-exact
  * features, eval-mode BatchNorm (randomised running statistics and affine parameters) and train-mode BatchNorm:
    fp32 tolerance stated per assert (the GPU conv is bit-exact per layer; BatchNorm statistics are reduced in a different
    order, and 20 layers of that compound)
  * gradients of the input features, every conv weight and every BatchNorm weight / bias: against float64 autograd over
    the same graph with the device's ReLU branch choices imposed (det_chain.forward_t64), relative to the largest element
    of each tensor
on the golden PassOccVox output of the reference (tests/golden/btc_small.npz) and on a full-size KITTI-shaped batch."""
import numpy as np
import pytest
import torch

import det_chain
from golden_batch import golden_batch

from btcdet_amd.config import load_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def M():
    from btcdet_amd.btc_path import BtcHotPath
    cfg = load_cfg()
    torch.manual_seed(0)
    model = BtcHotPath(cfg, device=DEV).to(DEV)
    bb = model.det_modules.backbone_3d
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():   # non-trivial BatchNorm state so that a wrong BN <-> conv pairing cannot pass
        for m in bb.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.bias.shape, generator=g))
    return cfg, model


def golden_inputs():
    g, scenes, bd = golden_batch()
    return g["occvfe_voxel_features"], g["occvfe_occ_voxel_features"], g["pov_voxel_coords"].astype(np.int32), 2


def full_size_inputs(model):
    """the detection backbone's inputs on a full-size synthetic KITTI batch (~28 k points per scene, det cap biting),
    produced by the occupancy branch + PassOccVox + OccVFE on the device"""
    import bench
    batch = bench.build_batches(1, 5, torch.device(DEV))[0]
    model.train()
    with torch.no_grad():
        bd = model.prepare(batch)
        bd, _, _, _ = model.forward_occ(bd)
        bd = model.det_modules.vfe(bd)
    return (bd["voxel_features"].float().cpu().numpy(), bd["occ_voxel_features"].float().cpu().numpy(),
            bd["voxel_coords"].int().cpu().numpy(), bd["batch_size"])


def run_gpu(model, feats, occ, coords, bs, train, with_grad=False, probes=None):
    bb, bev = model.det_modules.backbone_3d, model.det_modules.map_to_bev_module
    bb.train(train)
    x = torch.from_numpy(feats).to(DEV).requires_grad_(with_grad)
    d = {"voxel_features": x, "occ_voxel_features": torch.from_numpy(occ).to(DEV),
         "voxel_coords": torch.from_numpy(coords).to(DEV), "batch_size": bs}
    ctx = torch.enable_grad() if with_grad else torch.no_grad()
    with ctx:
        d = bev(bb(d))
        out, xc, sf = d["encoded_spconv_tensor"], d["multi_scale_3d_features"]["x_combine"], d["spatial_features"]
        if with_grad:
            bb.zero_grad()
            loss = (sf * probes[0]).sum() + (xc.features * probes[1]).sum()
            loss.backward()
            torch.cuda.synchronize()
    return out, xc, sf, x


def gpu_relu_masks(model, feats, occ, coords, bs):
    """{BatchNorm name: active units (N, C)} of a train-mode forward pass.  The stages normally run as compiled chains that
    never call the inner modules, so this pass runs layer by layer (spconv.modules.CHAIN_LAYERS = False): same kernels, same
    launch shapes, bit-identical activations (asserted by the caller on the final outputs)."""
    from btcdet_amd.spconv import modules as sp_modules
    bb = model.det_modules.backbone_3d
    masks, hooks = {}, []
    for name, m in bb.named_modules():
        if isinstance(m, sp_modules.SparseSequential) and len(m) == 3 and isinstance(m[1], torch.nn.BatchNorm1d):
            hooks.append(m.register_forward_hook(
                lambda mod, inp, out, name=name: masks.__setitem__(name + ".1", (out.features > 0).cpu().numpy())))
    saved = {k: v.clone() for k, v in bb.state_dict().items()}
    sp_modules.CHAIN_LAYERS = False
    try:
        out, xc, sf, _ = run_gpu(model, feats, occ, coords, bs, True)
    finally:
        sp_modules.CHAIN_LAYERS = True
        for h in hooks:
            h.remove()
        bb.load_state_dict(saved)
    return masks, out.features.clone(), xc.features.clone()


def rel_err(a, b):
    scale = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max()) / scale


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).astype(np.float64).ravel()) / max(np.linalg.norm(b.astype(np.float64).ravel()), 1e-30))


@pytest.mark.parametrize("which", ["golden", "full"])
def test_det_backbone_vs_oracle_chain(M, which):
    cfg, model = M
    feats, occ, coords, bs = golden_inputs() if which == "golden" else full_size_inputs(model)
    bb = model.det_modules.backbone_3d
    rb, lv = det_chain.geometry(coords)
    if which == "full":
        assert coords.shape[0] > 30000, coords.shape          # the 16 000-voxel cap per scene bites + added occupancy points
    sd = {k: v.detach().cpu().numpy() for k, v in bb.state_dict().items()}
    for train in (False, True):
        saved = {k: v.clone() for k, v in bb.state_dict().items()}
        out, xc, sf, _ = run_gpu(model, feats, occ, coords, bs, train)
        ref = det_chain.forward_np(sd, rb, lv, feats, occ, bs, train)
        # geometry: bit-exact
        np.testing.assert_array_equal(out.indices.cpu().numpy(), ref["out_indices"])
        np.testing.assert_array_equal(xc.indices.cpu().numpy(), ref["x_combine_indices"])
        assert [int(v) for v in out.spatial_shape] == ref["out_shape"] == [2, 200, 176]
        assert [int(v) for v in xc.spatial_shape] == ref["x_combine_shape"] == [5, 200, 176]
        assert tuple(sf.shape) == (bs, 256, 200, 176)
        # features
        errs = {"encoded_spconv_tensor": rel_err(out.features.cpu().numpy(), ref["out"]),
                "x_combine": rel_err(xc.features.cpu().numpy(), ref["x_combine"]),
                "spatial_features": rel_err(sf.cpu().numpy(), ref["spatial_features"])}
        print("det backbone vs oracle chain [%s, BN %s]: max |diff| / max |ref| = %s" % (which, "train" if train else "eval", errs))
        tol = 1e-6 if not train else 2e-5      # observed 1e-7 (eval) / 2e-6 (train: batch statistics reduced in another order)
        for k, e in errs.items():
            assert e < tol, (k, e)
        # zero pattern of the BEV map = the active set of the last level, exactly
        nz = (sf != 0).any(1).cpu().numpy()
        nz_ref = (ref["spatial_features"] != 0).any(1)
        assert np.array_equal(nz, nz_ref)
        bb.load_state_dict(saved)     # train mode moved the running statistics


@pytest.mark.parametrize("which", ["golden", "full"])
def test_det_backbone_gradients_vs_float64_autograd(M, which):
    cfg, model = M
    feats, occ, coords, bs = golden_inputs() if which == "golden" else full_size_inputs(model)
    bb = model.det_modules.backbone_3d
    rb, lv = det_chain.geometry(coords)
    sd = {k: v.detach().cpu().numpy() for k, v in bb.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    n_xc = rb["subm4"][0].shape[0]
    p_sf = torch.randn(bs, 256, 200, 176, generator=g)
    p_xc = torch.randn(n_xc, 128, generator=g)
    saved = {k: v.clone() for k, v in bb.state_dict().items()}
    masks, out0, xc0 = gpu_relu_masks(model, feats, occ, coords, bs)
    assert len(masks) == 18, sorted(masks)     # every conv -> BatchNorm -> ReLU block of the graph
    out, xc, sf, x = run_gpu(model, feats, occ, coords, bs, True, with_grad=True, probes=(p_sf.to(DEV), p_xc.to(DEV)))
    assert torch.equal(out.features, out0) and torch.equal(xc.features, xc0)     # the mask pass saw the same activations
    ref, params, x64 = det_chain.forward_t64(sd, rb, lv, feats, occ, bs, True, masks=masks)
    ((ref["spatial_features"] * p_sf.double()).sum() + (ref["x_combine"] * p_xc.double()).sum()).backward()
    got = {k: p.grad for k, p in bb.named_parameters()}
    missing = [k for k, p in bb.named_parameters() if p.requires_grad and p.grad is None]
    assert not missing, missing
    worst = {}
    for k, p64 in params.items():
        assert p64.grad is not None, k
        e = rel_err(got[k].float().cpu().numpy(), p64.grad.numpy())
        worst[k] = e
    gi, ri = x.grad.cpu().numpy(), x64.grad.numpy()
    e_in = rel_err(gi, ri)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print("det backbone grads vs float64 autograd [%s]: input %.2e, worst parameters %s" % (which, e_in, top))
    assert e_in < 2e-5, e_in            # observed 2e-6 (golden), 3e-6 (full size)
    for k, e in worst.items():
        assert e < 3e-5, (k, e)         # observed <= 3e-6
    bb.load_state_dict(saved)
    bb.zero_grad()
