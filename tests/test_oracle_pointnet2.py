"""CPU: the oracle's pointnet2_stack restatement (oracle/btc_oracle.c) against independent numpy
formulations of what the reference's CUDA kernels compute (ball_query_gpu.cu, shell_query_gpu.cu, group_points_gpu.cu,
sampling_gpu.cu, interpolate_gpu.cu).  The reference has no tests or vectors for these ops and they cannot run without CUDA:
parity is unpinned against an execution of the reference (oracle header)."""
import numpy as np
import pytest

from oracle import oracle as orc


def _scenes(rng, counts, lo=-3.0, hi=3.0):
    return rng.uniform(lo, hi, size=(int(sum(counts)), 3)).astype(np.float32), np.asarray(counts, dtype=np.int32)


def _d2(q, p):
    d = (q[None, :] - p).astype(np.float32)
    sq = (d * d).astype(np.float32)
    return ((sq[:, 0] + sq[:, 1]).astype(np.float32) + sq[:, 2]).astype(np.float32)


def _ball_query_np(radius, nsample, xyz, cnt, new_xyz, ncnt):
    inner, outer = (radius if isinstance(radius, (list, tuple)) else (None, radius))
    M = new_xyz.shape[0]
    idx = np.zeros((M, nsample), dtype=np.int32)
    empty = np.zeros((M,), dtype=bool)
    starts, nstarts = np.concatenate([[0], np.cumsum(cnt)]), np.concatenate([[0], np.cumsum(ncnt)])
    for b in range(len(cnt)):
        pts = xyz[starts[b]:starts[b + 1]]
        for q in range(nstarts[b], nstarts[b + 1]):
            d2 = _d2(new_xyz[q], pts) if pts.shape[0] else np.zeros((0,), np.float32)
            m = d2 < np.float32(outer) * np.float32(outer)
            if inner is not None:
                m &= d2 >= np.float32(inner) * np.float32(inner)
            hit = np.nonzero(m)[0][:nsample]
            if hit.size == 0:
                empty[q] = True
            else:
                idx[q, :] = hit[0]
                idx[q, :hit.size] = hit
    return idx, empty


@pytest.mark.parametrize("radius", [0.8, 2.5, [0.5, 1.5]])
def test_ball_and_shell_query(radius):
    rng = np.random.default_rng(1)
    xyz, cnt = _scenes(rng, [300, 0, 517])
    new_xyz, ncnt = _scenes(rng, [40, 7, 61], -4.0, 4.0)
    for nsample in (1, 16, 64):
        idx, empty = orc.ball_query(radius, nsample, xyz, cnt, new_xyz, ncnt)
        ridx, rempty = _ball_query_np(radius, nsample, xyz, cnt, new_xyz, ncnt)
        np.testing.assert_array_equal(idx, ridx)
        np.testing.assert_array_equal(empty, rempty)
    assert empty[40:47].all()          # queries of the empty scene
    assert 0 < empty.sum() < empty.size


def test_group_points_and_grad():
    rng = np.random.default_rng(2)
    feats, fc = rng.standard_normal((50 + 80, 5)).astype(np.float32), np.array([50, 80], np.int32)
    ic = np.array([9, 13], np.int32)
    idx = np.concatenate([rng.integers(0, 50, (9, 6)), rng.integers(0, 80, (13, 6))]).astype(np.int32)
    out = orc.group_points(feats, fc, idx, ic)
    gidx = idx + np.repeat([0, 50], ic)[:, None]
    np.testing.assert_array_equal(out, feats[gidx].transpose(0, 2, 1))
    g = rng.standard_normal(out.shape).astype(np.float32)
    ref = np.zeros((130, 5), np.float64)
    np.add.at(ref, gidx.reshape(-1), g.transpose(0, 2, 1).reshape(-1, 5).astype(np.float64))
    np.testing.assert_allclose(orc.group_points_grad(g, idx, ic, fc, 130), ref, rtol=1e-5, atol=1e-6)


def _fps_np(xyz, m):
    """argmax of the running min-distance with the tie rule the reference's thread layout implies: thread t = k mod T keeps its
    first maximum, and the pairwise tree (t, t + half), half = T/2 .. 1, lets the lower slot win ties -- so among equal maxima
    the winner has the smallest BIT-REVERSED thread index (the last merge separates even from odd t, the one before t mod 4
    in {0, 1} from {2, 3}, ...), then the smallest k; T = largest power of two <= n capped at 1024"""
    B, n, _ = xyz.shape
    T = 1
    while T * 2 <= n and T < 1024:
        T *= 2
    bits = T.bit_length() - 1
    rev = lambda t: int(format(t, "0%db" % bits)[::-1], 2) if bits else 0
    out = np.zeros((B, m), np.int32)
    for b in range(B):
        temp = np.full((n,), 1e10, np.float32)
        old = 0
        for j in range(1, m):
            d = (xyz[b] - xyz[b, old][None, :]).astype(np.float32)
            sq = (d * d).astype(np.float32)
            dd = ((sq[:, 0] + sq[:, 1]).astype(np.float32) + sq[:, 2]).astype(np.float32)
            temp = np.minimum(dd, temp)
            cand = np.nonzero(temp == temp.max())[0]
            old = int(min(cand, key=lambda k: (rev(k % T), k)))
            out[b, j] = old
    return out


@pytest.mark.parametrize("n,m", [(1, 1), (10, 6), (300, 40), (1500, 64), (2500, 33)])
def test_furthest_point_sampling_and_its_tie_rule(n, m):
    rng = np.random.default_rng(n)
    xyz = rng.uniform(-5, 5, (2, n, 3)).astype(np.float32)
    np.testing.assert_array_equal(orc.furthest_point_sample(xyz, m), _fps_np(xyz, m))
    lattice = rng.integers(0, 3, (2, n, 3)).astype(np.float32)          # many exactly equal distances
    np.testing.assert_array_equal(orc.furthest_point_sample(lattice, m), _fps_np(lattice, m))


def test_three_nn_and_interpolate():
    rng = np.random.default_rng(4)
    unknown, uc = _scenes(rng, [70, 33, 5])
    known, kc = _scenes(rng, [40, 2, 1])
    dist, idx = orc.three_nn(unknown, uc, known, kc)
    us, ks = np.concatenate([[0], np.cumsum(uc)]), np.concatenate([[0], np.cumsum(kc)])
    for b in range(3):
        pts = known[ks[b]:ks[b + 1]]
        for q in range(us[b], us[b + 1]):
            d2 = _d2(unknown[q], pts)
            order = np.argsort(d2, kind="stable")[:3]
            np.testing.assert_array_equal(idx[q, :order.size], order + ks[b])
            np.testing.assert_array_equal(dist[q, :order.size], np.sqrt(d2[order]))
            assert np.all(np.isinf(dist[q, order.size:])) and np.all(idx[q, order.size:] == ks[b])   # fewer than 3 known points
    feats = rng.standard_normal((43, 6)).astype(np.float32)
    w = rng.uniform(0, 1, idx.shape).astype(np.float32)
    out = orc.three_interpolate(feats, idx, w)
    ref = (w[:, 0:1] * feats[idx[:, 0]] + w[:, 1:2] * feats[idx[:, 1]]).astype(np.float32) + w[:, 2:3] * feats[idx[:, 2]]
    np.testing.assert_array_equal(out, ref.astype(np.float32))
    g = rng.standard_normal(out.shape).astype(np.float32)
    gref = np.zeros((43, 6), np.float64)
    for j in range(3):
        np.add.at(gref, idx[:, j], (g * w[:, j:j + 1]).astype(np.float64))
    np.testing.assert_allclose(orc.three_interpolate_grad(g, idx, w, 43), gref, rtol=1e-5, atol=1e-6)
