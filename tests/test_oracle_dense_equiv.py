"""Pins the CPU oracle's spconv restatement (parity otherwise unpinned, see oracle/btc_oracle.c)
against dense PyTorch ops: conv3d / conv_transpose3d / max_pool3d on the densified tensor, restricted
to the active output set -- the strongest spconv-independent check available."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as orc


def rand_indices(rng, n, batch, shape):
    vol = int(np.prod(shape))
    lin = rng.choice(batch * vol, size=min(n, batch * vol), replace=False)
    b, rem = lin // vol, lin % vol
    z, rem = rem // (shape[1] * shape[2]), rem % (shape[1] * shape[2])
    y, x = rem // shape[2], rem % shape[2]
    return np.stack([b, z, y, x], axis=1).astype(np.int32)


def densify(feat, idx, batch, shape):
    return torch.from_numpy(orc.dense(feat, idx, batch, shape))


def w_torch(W):  # [kD,kH,kW,Cin,Cout] -> [Cout,Cin,kD,kH,kW]
    return torch.from_numpy(W).permute(4, 3, 0, 1, 2).contiguous()


CASES = [
    # shape, k, s, p, d
    ((9, 15, 13), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    ((9, 15, 13), (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1)),
    ((11, 16, 12), (3, 3, 3), (2, 2, 2), (0, 1, 1), (1, 1, 1)),
    ((5, 10, 8), (3, 1, 1), (2, 1, 1), (0, 0, 0), (1, 1, 1)),
    ((2, 10, 8), (2, 1, 1), (2, 1, 1), (0, 0, 0), (1, 1, 1)),
    ((6, 12, 12), (3, 3, 3), (1, 2, 2), (1, 1, 1), (1, 1, 1)),
    ((8, 9, 7), (2, 2, 3), (1, 1, 1), (0, 0, 0), (1, 1, 1)),
    ((8, 9, 10), (3, 3, 3), (1, 1, 1), (2, 2, 2), (2, 2, 2)),
]


@pytest.mark.parametrize("shape,k,s,p,d", CASES)
def test_regular_conv_matches_dense(shape, k, s, p, d):
    rng = np.random.default_rng(0)
    B, cin, cout = 2, 5, 7
    idx = rand_indices(rng, 150, B, shape)
    feat = rng.standard_normal((idx.shape[0], cin)).astype(np.float32)
    W = rng.standard_normal((*k, cin, cout)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    out_idx, nbr_out, nbr_in, osh = orc.rulebook(idx, shape, k, s, p, d, orc.MODE_CONV)
    out = orc.conv_fwd(feat, W, bias, nbr_out)
    ref = F.conv3d(densify(feat, idx, B, shape).double(), w_torch(W).double(), torch.from_numpy(bias).double(),
                   stride=s, padding=p, dilation=d)
    assert tuple(ref.shape[2:]) == tuple(osh)
    # output rows sorted ascending in (b,z,y,x)
    lin = ((out_idx[:, 0].astype(np.int64) * osh[0] + out_idx[:, 1]) * osh[1] + out_idx[:, 2]) * osh[2] + out_idx[:, 3]
    assert np.all(np.diff(lin) > 0)
    got = ref[out_idx[:, 0], :, out_idx[:, 1], out_idx[:, 2], out_idx[:, 3]].numpy()
    np.testing.assert_allclose(out, got, rtol=1e-4, atol=1e-4)
    # the active output set is exactly the set reachable from an active input (dilation of the input set)
    reach = F.conv3d((densify(np.ones((idx.shape[0], 1), np.float32), idx, B, shape)),
                     torch.ones(1, 1, *k), stride=s, padding=p, dilation=d)[:, 0] > 0
    assert int(reach.sum()) == out_idx.shape[0]
    assert bool(reach[out_idx[:, 0], out_idx[:, 1], out_idx[:, 2], out_idx[:, 3]].all())
    # nbr_in is the transpose of nbr_out
    K = nbr_out.shape[1]
    for kk in range(K):
        o = np.nonzero(nbr_out[:, kk] >= 0)[0]
        assert np.array_equal(nbr_in[nbr_out[o, kk], kk], o)
        assert (nbr_in[:, kk] >= 0).sum() == o.size


@pytest.mark.parametrize("k,d", [((3, 3, 3), (1, 1, 1)), ((1, 3, 3), (1, 1, 1)), ((3, 3, 3), (1, 2, 2)), ((5, 3, 3), (1, 1, 1))])
def test_subm_matches_dense(k, d):
    rng = np.random.default_rng(1)
    shape, B, cin, cout = (7, 12, 11), 2, 4, 6
    idx = rand_indices(rng, 300, B, shape)
    feat = rng.standard_normal((idx.shape[0], cin)).astype(np.float32)
    W = rng.standard_normal((*k, cin, cout)).astype(np.float32)
    out_idx, nbr_out, nbr_in, osh = orc.rulebook(idx, shape, k, 1, 0, d, orc.MODE_SUBM)
    assert np.array_equal(out_idx, idx) and tuple(osh) == shape
    out = orc.conv_fwd(feat, W, None, nbr_out)
    pad = tuple((kk // 2) * dd for kk, dd in zip(k, d))
    ref = F.conv3d(densify(feat, idx, B, shape).double(), w_torch(W).double(), None, stride=1, padding=pad, dilation=d)
    got = ref[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]].numpy()
    np.testing.assert_allclose(out, got, rtol=1e-4, atol=1e-4)
    # centre offset is the identity
    kc = (((k[0] // 2) * k[1]) + k[1] // 2) * k[2] + k[2] // 2
    assert np.array_equal(nbr_out[:, kc], np.arange(idx.shape[0]))


@pytest.mark.parametrize("shape,k,s,p", [((3, 8, 7), (3, 3, 3), (2, 2, 2), (1, 1, 1)), ((5, 6, 6), (3, 3, 3), (2, 2, 2), (1, 1, 1)),
                                         ((4, 5, 6), (2, 2, 2), (2, 2, 2), (0, 0, 0))])
def test_transpose_conv_matches_dense(shape, k, s, p):
    rng = np.random.default_rng(2)
    B, cin, cout = 2, 3, 5
    idx = rand_indices(rng, 60, B, shape)
    feat = rng.standard_normal((idx.shape[0], cin)).astype(np.float32)
    W = rng.standard_normal((*k, cin, cout)).astype(np.float32)
    out_idx, nbr_out, nbr_in, osh = orc.rulebook(idx, shape, k, s, p, 1, orc.MODE_TRANSPOSE)
    out = orc.conv_fwd(feat, W, None, nbr_out)
    wt = torch.from_numpy(W).permute(3, 4, 0, 1, 2).contiguous()  # [Cin,Cout,kD,kH,kW]
    ref = F.conv_transpose3d(densify(feat, idx, B, shape).double(), wt.double(), None, stride=s, padding=p)
    assert tuple(ref.shape[2:]) == tuple(osh)
    got = ref[out_idx[:, 0], :, out_idx[:, 1], out_idx[:, 2], out_idx[:, 3]].numpy()
    np.testing.assert_allclose(out, got, rtol=1e-4, atol=1e-4)
    # everything outside the active output set is exactly zero in the dense result
    mask = torch.ones_like(ref[:, 0], dtype=torch.bool)
    mask[out_idx[:, 0], out_idx[:, 1], out_idx[:, 2], out_idx[:, 3]] = False
    assert float(ref.abs().sum(1)[mask].max()) == 0.0


def test_maxpool_matches_dense_nonneg():
    rng = np.random.default_rng(3)
    shape, B, C = (9, 14, 12), 2, 2
    idx = rand_indices(rng, 400, B, shape)
    feat = rng.random((idx.shape[0], C)).astype(np.float32)  # >= 0 (occupancy code channels are probabilities/flags)
    out_idx, nbr_out, nbr_in, osh = orc.rulebook(idx, shape, 3, 2, 1, 1, orc.MODE_CONV)
    out = orc.maxpool_fwd(feat, nbr_out)
    ref = F.max_pool3d(densify(feat, idx, B, shape), 3, 2, 1)
    got = ref[out_idx[:, 0], :, out_idx[:, 1], out_idx[:, 2], out_idx[:, 3]].numpy()
    np.testing.assert_array_equal(out, got)
    # backward: compare with autograd through max_pool3d for tie-free data
    dout = rng.standard_normal(out.shape).astype(np.float32)
    din = orc.maxpool_bwd(feat, out, dout, nbr_in)
    x = densify(feat, idx, B, shape).requires_grad_(True)
    y = F.max_pool3d(x, 3, 2, 1)
    g = torch.zeros_like(y)
    g[out_idx[:, 0], :, out_idx[:, 1], out_idx[:, 2], out_idx[:, 3]] = torch.from_numpy(dout)
    y.backward(g)
    got = x.grad[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]].numpy()
    np.testing.assert_allclose(din, got, rtol=1e-6, atol=1e-6)


def test_conv_backward_matches_autograd():
    rng = np.random.default_rng(4)
    shape, B, cin, cout, k, s, p = (7, 10, 9), 2, 4, 5, (3, 3, 3), (2, 2, 2), (1, 1, 1)
    idx = rand_indices(rng, 200, B, shape)
    feat = rng.standard_normal((idx.shape[0], cin)).astype(np.float32)
    W = rng.standard_normal((*k, cin, cout)).astype(np.float32)
    out_idx, nbr_out, nbr_in, osh = orc.rulebook(idx, shape, k, s, p, 1, orc.MODE_CONV)
    dout = rng.standard_normal((out_idx.shape[0], cout)).astype(np.float32)
    din = orc.conv_dgrad(dout, W, nbr_in)
    dW = orc.conv_wgrad(feat, dout, nbr_out, W.shape)
    x = densify(feat, idx, B, shape).double().requires_grad_(True)
    wt = w_torch(W).double().requires_grad_(True)
    y = F.conv3d(x, wt, None, stride=s, padding=p)
    g = torch.zeros_like(y)
    g[out_idx[:, 0], :, out_idx[:, 1], out_idx[:, 2], out_idx[:, 3]] = torch.from_numpy(dout).double()
    y.backward(g)
    np.testing.assert_allclose(din, x.grad[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]].numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dW, wt.grad.permute(2, 3, 4, 1, 0).numpy(), rtol=1e-4, atol=1e-4)


def test_btcdet_resolution_ladders():
    """SURVEY.md App. A.2 (shape invariants the reference relies on)."""
    assert list(orc.out_shape([41, 1600, 1408], 3, 2, 1, 1, orc.MODE_CONV)) == [21, 800, 704]
    assert list(orc.out_shape([21, 800, 704], 3, 2, 1, 1, orc.MODE_CONV)) == [11, 400, 352]
    assert list(orc.out_shape([11, 400, 352], 3, 2, (0, 1, 1), 1, orc.MODE_CONV)) == [5, 200, 176]
    assert list(orc.out_shape([5, 200, 176], (3, 1, 1), (2, 1, 1), 0, 1, orc.MODE_CONV)) == [2, 200, 176]
    assert list(orc.out_shape([2, 200, 176], (2, 1, 1), (2, 1, 1), 0, 1, orc.MODE_CONV)) == [1, 200, 176]
    assert list(orc.out_shape([9, 157, 209], 3, 1, 1, 1, orc.MODE_CONV)) == [9, 157, 209]
    assert list(orc.out_shape([9, 157, 209], 3, 2, 1, 1, orc.MODE_CONV)) == [5, 79, 105]
    assert list(orc.out_shape([5, 79, 105], 3, 2, 1, 1, orc.MODE_CONV)) == [3, 40, 53]
    assert list(orc.out_shape([3, 40, 53], 3, 2, 1, 1, orc.MODE_TRANSPOSE)) == [5, 79, 105]
    assert list(orc.out_shape([5, 79, 105], 3, 2, 1, 1, orc.MODE_TRANSPOSE)) == [9, 157, 209]


def test_voxelizer_known_answer():
    """Hand-computed first-come voxelization (SURVEY.md App. B.1 semantics)."""
    g = orc.VoxelGeneratorV2([1.0, 1.0, 1.0], [0, 0, 0, 4, 3, 2], max_num_points=2, max_voxels=3)
    assert list(g.grid_size) == [4, 3, 2]
    pts = np.array([
        [0.5, 0.5, 0.5, 10],   # voxel A (z0,y0,x0) first
        [3.5, 2.5, 1.5, 11],   # voxel B (1,2,3)
        [0.6, 0.4, 0.1, 12],   # A second point
        [0.7, 0.7, 0.7, 13],   # A third -> dropped (max_num_points=2)
        [4.0, 0.0, 0.0, 14],   # x == upper bound -> out of range (half-open)
        [-0.01, 0.0, 0.0, 15], # below range
        [1.5, 0.5, 0.5, 16],   # voxel C (0,0,1)
        [2.5, 0.5, 0.5, 17],   # voxel D -> dropped (max_voxels=3)
        [1.2, 0.2, 0.2, 18],   # C second point
        [2.6, 0.5, 0.5, 19],   # D again -> still dropped
    ], dtype=np.float32)
    r = g.generate(pts)
    assert r["voxel_num"] == 3
    np.testing.assert_array_equal(r["coordinates"], [[0, 0, 0], [1, 2, 3], [0, 0, 1]])
    np.testing.assert_array_equal(r["num_points_per_voxel"], [2, 1, 2])
    np.testing.assert_array_equal(r["voxels"][0], pts[[0, 2]])
    np.testing.assert_array_equal(r["voxels"][1], [pts[1], np.zeros(4)])
    np.testing.assert_array_equal(r["voxels"][2], pts[[6, 8]])
    # scratch restored: a second call gives the same answer
    r2 = g.generate(pts)
    np.testing.assert_array_equal(r2["coordinates"], r["coordinates"])
