"""The backbone variants the reference reaches through BACKBONE_3D.NAME on the HIP operator set:
residual blocks, lateral-merge decoders, SparseInverseConv3d decoders -- pinned against the reference's own classes
(tests/golden/variants.npz), plus the residual block against dense torch ops."""
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_hip_core import dev, rand_indices

pytestmark = pytest.mark.gpu


def _batch(rng, n, B, shape, c):
    idx = rand_indices(rng, n, B, shape)
    idx = idx[np.lexsort((idx[:, 3], idx[:, 2], idx[:, 1], idx[:, 0]))]
    feats = rng.standard_normal((idx.shape[0], c)).astype(np.float32)
    return {"voxel_features": torch.from_numpy(feats).to(dev()), "voxel_coords": torch.from_numpy(idx).to(dev()), "batch_size": B}


def _golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "variants.npz"))


@pytest.mark.parametrize("name,which", [("VoxelBackBoneDeconvRes", "occ"), ("VoxelBackBoneInverseRes", "occ"), ("VoxelResBackBone8x", "det")])
def test_backbone_variants_vs_reference_modules(name, which):
    """the three variants against the REFERENCE's own classes (spconv_backbone.py:226-627) executed over the oracle-backed spconv
    (tests/golden/gen_variants_golden.py -> variants.npz): full-size synthetic KITTI input on the occupancy / detection grid,
    name-keyed weights, train- and eval-mode BatchNorm -- output indices bit-exact (SHA-1), features within 2e-5 of the scale,
    every multi-scale tensor of the residual 8x backbone included; then one backward pass (finite gradients everywhere)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import common
    from btcdet_amd import backbones_3d
    from btcdet_amd.config import load_cfg
    g = _golden()
    feats, coords, grid, B = common.variant_inputs(which)
    assert np.array_equal(common.sha1(coords), g[name + "_in_coords_sha1"]) and np.array_equal(common.sha1(feats), g[name + "_in_feats_sha1"])
    cfg = load_cfg()
    model_cfg = cfg.MODEL.OCC.BACKBONE_3D if which == "occ" else cfg.MODEL.BACKBONE_3D
    net = backbones_3d.__all__[name](model_cfg=model_cfg, input_channels=4, grid_size=np.array(grid))
    common.init_by_name(net)
    net = net.to(dev())
    bd_in = {"voxel_features": torch.from_numpy(feats).to(dev()), "voxel_coords": torch.from_numpy(coords).to(dev()).float(), "batch_size": B}
    for mode in ("train", "eval"):
        net.train(mode == "train")
        state = {k: v.clone() for k, v in net.state_dict().items()}
        with torch.no_grad():
            bd = net(dict(bd_in))
        net.load_state_dict(state)
        outs = {"out": bd["encoded_spconv_tensor"]}
        outs.update(bd.get("multi_scale_3d_features") or {})
        keys = [k[len(name) + len(mode) + 2:-len("_n")] for k in g.files if k.startswith("%s_%s_" % (name, mode)) and k.endswith("_n")]
        assert sorted(keys) == sorted(outs), (keys, sorted(outs))
        for key, x in outs.items():
            p = "%s_%s_%s_" % (name, mode, key)
            assert x.features.shape[0] == int(g[p + "n"]) and [int(v) for v in x.spatial_shape] == list(g[p + "shape"]), (p, x.features.shape)
            assert np.array_equal(common.sha1(x.indices.cpu().numpy().astype(np.int32)), g[p + "indices_sha1"]), p
            scale = float(np.abs(g[p + "features__sample"]).max())
            err, _ = common.check_digest(g, p + "features", x.features.float().cpu().numpy(), rtol=0, atol=2e-5 * scale, what=p)
            print("%s %s %s: max |diff| %.2e of scale %.2e" % (name, mode, key, err, scale))
    net.train()
    out = net(dict(bd_in))["encoded_spconv_tensor"]
    out.features.pow(2).mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


def test_sparse_basic_block_vs_dense_torch():
    """SubM conv = dense conv3d evaluated at the active cells of a zero-filled volume; BatchNorm over the active rows"""
    from btcdet_amd import backbones_3d
    from btcdet_amd import spconv
    rng = np.random.default_rng(5)
    B, shape, C = 2, (6, 14, 12), 16
    bd = _batch(rng, 500, B, shape, C)
    norm_fn = partial(torch.nn.BatchNorm1d, eps=1e-3, momentum=0.01)
    torch.manual_seed(1)
    blk = backbones_3d.SparseBasicBlock(C, C, norm_fn=norm_fn, indice_key="res").to(dev()).train()
    with torch.no_grad():
        for bn in (blk.bn1, blk.bn2):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
    x = spconv.SparseConvTensor(bd["voxel_features"].clone().requires_grad_(True), bd["voxel_coords"], list(shape), B)
    feats_in = x.features
    y = blk(x)
    y.features.pow(2).sum().backward()

    idx = bd["voxel_coords"].long()
    f_ref = bd["voxel_features"].clone().requires_grad_(True)

    def subm(feats, conv):
        dense = torch.zeros((B, feats.shape[1]) + shape, device=dev())
        dense = dense.index_put((idx[:, 0], slice(None), idx[:, 1], idx[:, 2], idx[:, 3]), feats) if False else dense
        dense = dense.permute(0, 2, 3, 4, 1).index_put((idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]), feats).permute(0, 4, 1, 2, 3)
        w = conv.weight.permute(4, 3, 0, 1, 2)  # [kD,kH,kW,Cin,Cout] -> [Cout,Cin,kD,kH,kW]
        out = F.conv3d(dense, w, conv.bias, padding=1)
        return out.permute(0, 2, 3, 4, 1)[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]]

    def bn(feats, m):
        return F.batch_norm(feats, None, None, m.weight, m.bias, True, 0.0, m.eps)

    h = torch.relu(bn(subm(f_ref, blk.conv1), blk.bn1))
    ref = torch.relu(bn(subm(h, blk.conv2), blk.bn2) + f_ref)
    ref.pow(2).sum().backward()
    torch.testing.assert_close(y.features, ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(feats_in.grad, f_ref.grad, rtol=1e-3, atol=1e-3)
