"""The backbone variants the reference reaches through BACKBONE_3D.NAME on the HIP operator set:
residual blocks, lateral-merge decoders, SparseInverseConv3d decoders."""
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


@pytest.mark.parametrize("name", ["VoxelBackBoneDeconvRes", "VoxelBackBoneInverseRes"])
def test_res_decoder_backbones_run(name):
    from btcdet_amd import backbones_3d
    from btcdet_amd.config import load_cfg
    rng = np.random.default_rng(3)
    grid = np.array([45, 37, 9])                       # 4m+1 per axis: stride-2 / transposed pairs round-trip
    bd = _batch(rng, 1500, 2, (9, 37, 45), 4)
    torch.manual_seed(0)
    net = backbones_3d.__all__[name](model_cfg=load_cfg().MODEL.OCC.BACKBONE_3D, input_channels=4, grid_size=grid).to(dev()).train()
    out = net(dict(bd))["encoded_spconv_tensor"]
    assert out.features.shape[1] == net.num_point_features == 32 and list(out.spatial_shape) == [9, 37, 45]
    if name == "VoxelBackBoneInverseRes":              # inverse convs restore the encoder's active set exactly
        assert torch.equal(out.indices, bd["voxel_coords"])
    else:                                              # conv1 is a dilating SparseConv3d: a superset of the input cells
        assert out.features.shape[0] > bd["voxel_coords"].shape[0]
    out.features.pow(2).mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


def test_res_backbone_8x_runs():
    from btcdet_amd import backbones_3d
    from btcdet_amd.config import load_cfg
    rng = np.random.default_rng(4)
    bd = _batch(rng, 4000, 2, (41, 96, 88), 4)
    torch.manual_seed(0)
    net = backbones_3d.VoxelResBackBone8x(model_cfg=load_cfg().MODEL.BACKBONE_3D, input_channels=4, grid_size=np.array([88, 96, 40])).to(dev())
    net.train()
    ret = net(dict(bd))
    out = ret["encoded_spconv_tensor"]
    assert list(out.spatial_shape) == [2, 12, 11] and out.features.shape[1] == 128 and ret["encoded_spconv_tensor_stride"] == 8
    assert [ret["multi_scale_3d_features"][k].features.shape[1] for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4")] == [16, 32, 64, 128]
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
