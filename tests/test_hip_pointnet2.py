"""pointnet2_stack HIP kernels (csrc/pointnet2.hip) against the C oracle: indices bit-exact (ball / shell
query, furthest point sampling incl. its tie rule, three-NN), gathers and the interpolation forward bit-exact, gradients (float
atomics in the reference and here) within 1e-5; at the ROI head's sizes (2 x 128 rois x 6^3 grid points against ~20 K points
per scene, conv_head.py:262-300) and on the edge cases the kernels branch on."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _scenes(rng, counts, lo, hi):
    return rng.uniform(lo, hi, size=(int(sum(counts)), 3)).astype(np.float32), np.asarray(counts, dtype=np.int32)


@pytest.mark.parametrize("case", ["roi_grid", "empty_scene", "tiny", "shell", "straddle"])
def test_ball_query_bit_exact(case):
    from btcdet_amd import pointnet2_stack as p2
    rng = np.random.default_rng(3)
    radius, nsample = 0.8, 16
    if case == "roi_grid":
        xyz, cnt = _scenes(rng, [19000, 21500], [0, -40, -3], [70, 40, 1])
        centers = xyz[rng.integers(0, xyz.shape[0], 256)]
        grid = (centers[:, None, :] + rng.uniform(-2, 2, (256, 216, 3))).reshape(-1, 3).astype(np.float32)
        new_xyz, ncnt = grid, np.array([128 * 216, 128 * 216], np.int32)
    elif case == "empty_scene":
        xyz, cnt = _scenes(rng, [700, 0, 1300], -3, 3)
        new_xyz, ncnt = _scenes(rng, [50, 9, 77], -4, 4)
    elif case == "tiny":
        xyz, cnt = _scenes(rng, [3], -1, 1)
        new_xyz, ncnt = _scenes(rng, [5], -1, 1)
        radius, nsample = 1.5, 8
    elif case == "shell":
        xyz, cnt = _scenes(rng, [5000, 4000], -3, 3)
        new_xyz, ncnt = _scenes(rng, [333, 444], -3, 3)
        radius, nsample = [0.4, 1.2], 32
    else:   # scene boundary inside a 16-query workgroup, more samples than a wave has lanes
        xyz, cnt = _scenes(rng, [2100, 1900, 2300], -2, 2)
        new_xyz, ncnt = _scenes(rng, [7, 5, 41], -2, 2)
        radius, nsample = 1.0, 100
    idx, empty = p2.ball_query(radius, nsample, t(xyz), t(cnt), t(new_xyz), t(ncnt))
    ridx, rempty = orc.ball_query(radius, nsample, xyz, cnt, new_xyz, ncnt)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(empty.cpu().numpy(), rempty)
    assert idx.dtype == torch.int32 and empty.dtype == torch.bool


def test_grouping_forward_exact_backward_close():
    from btcdet_amd import pointnet2_stack as p2
    rng = np.random.default_rng(5)
    fc, ic = np.array([6000, 4500], np.int32), np.array([900, 1100], np.int32)
    feats = rng.standard_normal((int(fc.sum()), 13)).astype(np.float32)
    idx = np.concatenate([rng.integers(0, 6000, (900, 16)), rng.integers(0, 4500, (1100, 16))]).astype(np.int32)
    f = t(feats).requires_grad_(True)
    out = p2.grouping_operation(f, t(fc), t(idx), t(ic))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), orc.group_points(feats, fc, idx, ic))
    g = rng.standard_normal(tuple(out.shape)).astype(np.float32)
    out.backward(t(g))
    ref = orc.group_points_grad(g, idx, ic, fc, feats.shape[0])
    np.testing.assert_allclose(f.grad.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n,m,lattice", [(1, 1, False), (10, 6, True), (1500, 64, True), (4096, 512, False), (16384, 2048, False),
                                         (5000, 100, True), (20000, 300, False)])
def test_furthest_point_sampling_bit_exact(n, m, lattice):
    from btcdet_amd import pointnet2_stack as p2
    rng = np.random.default_rng(n)
    xyz = (rng.integers(0, 4, (2, n, 3)) if lattice else rng.uniform(-30, 30, (2, n, 3))).astype(np.float32)
    got = p2.furthest_point_sample(t(xyz), m)
    np.testing.assert_array_equal(got.cpu().numpy(), orc.furthest_point_sample(xyz, m))


def test_three_nn_and_interpolate():
    from btcdet_amd import pointnet2_stack as p2
    rng = np.random.default_rng(7)
    unknown, uc = _scenes(rng, [3000, 2, 2500], -5, 5)
    known, kc = _scenes(rng, [1200, 1, 2100], -5, 5)
    dist, idx = p2.three_nn(t(unknown), t(uc), t(known), t(kc))
    rdist, ridx = orc.three_nn(unknown, uc, known, kc)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(dist.cpu().numpy(), rdist)
    feats = rng.standard_normal((known.shape[0], 32)).astype(np.float32)
    w = rng.uniform(0, 1, ridx.shape).astype(np.float32)
    f = t(feats).requires_grad_(True)
    out = p2.three_interpolate(f, idx, t(w))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), orc.three_interpolate(feats, ridx, w))
    g = rng.standard_normal(tuple(out.shape)).astype(np.float32)
    out.backward(t(g))
    np.testing.assert_allclose(f.grad.cpu().numpy(), orc.three_interpolate_grad(g, ridx, w, feats.shape[0]), rtol=1e-5, atol=1e-5)


def test_stack_sa_module_msg_matches_a_torch_formulation_on_oracle_indices():
    """StackSAModuleMSG (ROI-head configuration: two radii, rotation and scaling of the grouped offsets) against the same
    module arithmetic evaluated in torch from the ORACLE's ball-query indices"""
    from btcdet_amd import pointnet2_stack as p2
    rng = np.random.default_rng(9)
    torch.manual_seed(0)
    xyz, cnt = _scenes(rng, [4000, 3500], [0, -10, -2], [20, 10, 1])
    feats = rng.standard_normal((xyz.shape[0], 1)).astype(np.float32)
    n_roi, G = 6, 27
    new_xyz = (xyz[rng.integers(0, xyz.shape[0], 2 * n_roi)][:, None, :] + rng.uniform(-1, 1, (2 * n_roi, G, 3))).reshape(-1, 3).astype(np.float32)
    ncnt = np.array([n_roi * G, n_roi * G], np.int32)
    yaw = rng.uniform(-3, 3, 2 * n_roi).astype(np.float32)
    rot = np.zeros((2 * n_roi, 3, 3), np.float32)
    rot[:, 0, 0], rot[:, 0, 1], rot[:, 1, 0], rot[:, 1, 1], rot[:, 2, 2] = np.cos(yaw), -np.sin(yaw), np.sin(yaw), np.cos(yaw), 1
    xys = np.repeat(rng.uniform(1, 3, 2 * n_roi).astype(np.float32), G).reshape(-1, 1, 1)
    zs = np.repeat(rng.uniform(1, 2, 2 * n_roi).astype(np.float32), G).reshape(-1, 1, 1)
    mod = p2.StackSAModuleMSG(radii=[0.8, 1.6], nsamples=[16, 16], mlps=[[1, 8, 8], [1, 8, 8]], use_xyz=True, pool_method='max_pool').to(DEV).eval()
    _, got = mod(t(xyz), t(cnt), t(new_xyz), t(ncnt), t(feats), rotateMatrix=t(rot), xyscales=t(xys), zscales=t(zs))
    outs = []
    starts = np.repeat(np.concatenate([[0], np.cumsum(cnt)[:-1]]), ncnt)
    for k, r in enumerate([0.8, 1.6]):
        idx, empty = orc.ball_query(r, 16, xyz, cnt, new_xyz, ncnt)
        gidx = t((idx + starts[:, None]).astype(np.int64))
        gx = t(xyz)[gidx].permute(0, 2, 1) - t(new_xyz).unsqueeze(-1)
        gx[t(empty)] = 0
        R = t(rot).view(2 * n_roi, 1, 3, 3).repeat(1, G, 1, 1).view(-1, 3, 3)
        gx = torch.einsum("nmj,nij->nmi", gx.permute(0, 2, 1), R).permute(0, 2, 1)
        gx = torch.cat([gx[:, :2] / t(xys), gx[:, 2:3] / t(zs)], dim=1)
        gf = t(feats)[gidx].permute(0, 2, 1).clone()
        gf[t(empty)] = 0
        x = torch.cat([gx, gf], dim=1).permute(1, 0, 2).unsqueeze(0)
        x = mod.mlps[k](x)
        outs.append(torch.nn.functional.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1).squeeze(0).permute(1, 0))
    ref = torch.cat(outs, dim=1)
    assert got.shape == (2 * n_roi * G, 16)
    torch.testing.assert_close(got, ref, rtol=1e-6, atol=1e-6)
