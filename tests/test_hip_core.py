"""GPU parity tests of the HIP hot path against the CPU oracle, through the C ABI (ctypes).
Bit-exact for voxel indices / rulebooks / fwd+dgrad features (identical fmaf chain); stated fp32
tolerance for wgrad (split-K partial sums)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rand_indices(rng, n, batch, shape):
    vol = int(np.prod(shape))
    lin = rng.choice(batch * vol, size=min(n, batch * vol), replace=False)
    b, rem = lin // vol, lin % vol
    z, rem = rem // (shape[1] * shape[2]), rem % (shape[1] * shape[2])
    y, x = rem // shape[2], rem % shape[2]
    return np.stack([b, z, y, x], axis=1).astype(np.int32)


# ------------------------------------------------------------------ voxelizer
def _check_voxelizer(points_list, vsize, rng_range, max_pts, max_vox):
    from btcdet_amd.spconv import utils
    gen = utils.VoxelGeneratorV2(vsize, rng_range, max_pts, max_vox)
    ogen = orc.VoxelGeneratorV2(vsize, rng_range, max_pts, max_vox)
    assert list(gen.grid_size) == list(ogen.grid_size)
    pts = np.concatenate(points_list, axis=0).astype(np.float32)
    offs = np.cumsum([0] + [p.shape[0] for p in points_list]).astype(np.int32)
    v, c, n = gen.generate_batch(torch.from_numpy(pts).to(dev()), torch.from_numpy(offs).to(dev()))
    v, c, n = v.cpu().numpy(), c.cpu().numpy(), n.cpu().numpy()
    row = 0
    for b, p in enumerate(points_list):
        r = ogen.generate(p)
        m = r["voxel_num"]
        np.testing.assert_array_equal(c[row:row + m, 0], b)
        np.testing.assert_array_equal(c[row:row + m, 1:], r["coordinates"])
        np.testing.assert_array_equal(n[row:row + m], r["num_points_per_voxel"])
        np.testing.assert_array_equal(v[row:row + m], r["voxels"])
        row += m
    assert row == v.shape[0]


def test_voxelizer_known_answer_and_caps():
    pts = np.array([[0.5, 0.5, 0.5, 10], [3.5, 2.5, 1.5, 11], [0.6, 0.4, 0.1, 12], [0.7, 0.7, 0.7, 13], [4.0, 0.0, 0.0, 14],
                    [-0.01, 0.0, 0.0, 15], [1.5, 0.5, 0.5, 16], [2.5, 0.5, 0.5, 17], [1.2, 0.2, 0.2, 18],
                    [2.6, 0.5, 0.5, 19]], dtype=np.float32)
    _check_voxelizer([pts], [1.0, 1.0, 1.0], [0, 0, 0, 4, 3, 2], 2, 3)
    _check_voxelizer([pts, pts[::-1].copy(), pts[:0]], [1.0, 1.0, 1.0], [0, 0, 0, 4, 3, 2], 2, 3)


@pytest.mark.parametrize("seed", [0, 1])
def test_voxelizer_random_dense_cells(seed):
    rng = np.random.default_rng(seed)
    # many points per cell (exercises the atomicMin cascade and the max_points cut) and a tight voxel cap
    scenes = [rng.uniform(-0.5, 10.5, (n, 5)).astype(np.float32) for n in (5000, 1, 3000)]
    _check_voxelizer(scenes, [1.0, 0.5, 2.0], [0, 0, 0, 10, 10, 10], 7, 150)
    _check_voxelizer(scenes, [0.25, 0.25, 0.25], [0, 0, 0, 10, 10, 10], 3, 100000)


def test_voxelizer_kitti_scene_both_grids():
    from btcdet_amd import synth
    b = synth.make_batch([1000, 1001])
    scenes = [s["points"] for s in b["scenes"]]
    _check_voxelizer(scenes, synth.KITTI_DET_VOXEL, synth.KITTI_DET_RANGE, 5, 16000)
    cyl = [orc.absxyz_2_cylinxyz_np(s["pre_rot_points"]) for s in b["scenes"]]
    _check_voxelizer(cyl, synth.KITTI_OCC_VOXEL, synth.KITTI_OCC_RANGE, 12, 20000)


def test_cart_to_cylinder_and_sphere():
    from btcdet_amd import _lib, synth
    p = synth.make_scene(7)["points"]
    x = torch.from_numpy(p).to(dev())
    for mode, fn in ((1, orc.absxyz_2_cylinxyz_np), (2, orc.absxyz_2_spherexyz_np)):
        out = torch.empty_like(x)
        _lib.check(_lib.lib().btc_cart_to_occ_coords(_lib.ptr(x), _lib.ptr(out), x.shape[0], x.shape[1], mode,
                                                     _lib.stream_ptr()), "cart")
        ref = fn(p)
        # fp32 transcendental (atan2f device vs libm): tolerance 2e-6 relative / 2e-5 deg absolute
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-6, atol=2e-5)
        np.testing.assert_array_equal(out.cpu().numpy()[:, 3], p[:, 3])


# ------------------------------------------------------------------ rulebook
RB_CASES = [
    ((9, 15, 13), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), "conv"),
    ((9, 15, 13), (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), "conv"),
    ((11, 16, 12), (3, 3, 3), (2, 2, 2), (0, 1, 1), (1, 1, 1), "conv"),
    ((5, 10, 8), (3, 1, 1), (2, 1, 1), (0, 0, 0), (1, 1, 1), "conv"),
    ((2, 10, 8), (2, 1, 1), (2, 1, 1), (0, 0, 0), (1, 1, 1), "conv"),
    ((6, 12, 12), (3, 3, 3), (1, 2, 2), (1, 1, 1), (1, 1, 1), "conv"),
    ((8, 9, 10), (3, 3, 3), (1, 1, 1), (2, 2, 2), (2, 2, 2), "conv"),
    ((3, 8, 7), (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), "transpose"),
    ((4, 5, 6), (2, 2, 2), (2, 2, 2), (0, 0, 0), (1, 1, 1), "transpose"),
    ((7, 12, 11), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), "subm"),
    ((7, 12, 11), (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), "subm"),
    ((7, 12, 11), (3, 3, 3), (1, 1, 1), (0, 0, 0), (1, 2, 2), "subm"),
]


def _rb_both(idx, B, shape, k, s, p, d, kind):
    from btcdet_amd.spconv import ops
    mode = {"conv": orc.MODE_CONV, "transpose": orc.MODE_TRANSPOSE, "subm": orc.MODE_SUBM}[kind]
    o_idx, o_out, o_in, o_sh = orc.rulebook(idx, shape, k, s, p, d, mode)
    rb = ops.build_rulebook(torch.from_numpy(idx).to(dev()), B, shape, k, s, p, d, 0, kind == "subm", kind == "transpose")
    return (o_idx, o_out, o_in, o_sh), rb


@pytest.mark.parametrize("shape,k,s,p,d,kind", RB_CASES)
def test_rulebook_bit_exact(shape, k, s, p, d, kind):
    rng = np.random.default_rng(5)
    B = 3
    idx = rand_indices(rng, 400, B, shape)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, k, s, p, d, kind)
    assert list(rb.out_shape) == list(o_sh)
    np.testing.assert_array_equal(rb.out_indices.cpu().numpy(), o_idx)
    np.testing.assert_array_equal(rb.nbr_out.cpu().numpy(), o_out)
    np.testing.assert_array_equal(rb.nbr_in.cpu().numpy(), o_in)
    # spconv-layout view == canonical pair lists
    pairs, num = rb.indice_pairs()
    pairs, num = pairs.cpu().numpy(), num.cpu().numpy()
    cp, cn = orc.canonical_pairs(o_out)
    np.testing.assert_array_equal(num, cn)
    for kk in range(len(cp)):
        np.testing.assert_array_equal(pairs[:, kk, :cn[kk]], cp[kk])
        assert np.all(pairs[:, kk, cn[kk]:] == -1)


def test_rulebook_empty_and_single():
    from btcdet_amd.spconv import ops
    for n in (0, 1):
        idx = np.array([[0, 1, 2, 3]], np.int32)[:n]
        for kind in ("subm", "conv"):
            rb = ops.build_rulebook(torch.from_numpy(idx).to(dev()).reshape(-1, 4), 1, (4, 6, 8), 3, 2 if kind == "conv" else 1, 1, 1, 0,
                                    kind == "subm", False)
            o_idx, o_out, o_in, _ = orc.rulebook(idx.reshape(-1, 4), (4, 6, 8), 3, 2 if kind == "conv" else 1, 1, 1,
                                                  orc.MODE_SUBM if kind == "subm" else orc.MODE_CONV)
            np.testing.assert_array_equal(rb.out_indices.cpu().numpy(), o_idx)
            np.testing.assert_array_equal(rb.nbr_out.cpu().numpy(), o_out)


def test_rulebook_kitti_level_shapes_and_row_alignment():
    """conv2 and the max-pool on the same geometry must emit identical output rows (sparse_cat,
    spconv_backbone.py:869-873,972-974); both equal the oracle."""
    from btcdet_amd import synth
    from btcdet_amd.spconv import ops
    b = synth.make_batch([1000, 1001])
    og = orc.VoxelGeneratorV2(synth.KITTI_DET_VOXEL, synth.KITTI_DET_RANGE, 5, 16000)
    idx = np.concatenate([np.pad(og.generate(s["points"])["coordinates"], ((0, 0), (1, 0)), constant_values=i)
                          for i, s in enumerate(b["scenes"])]).astype(np.int32)
    shape = [41, 1600, 1408]
    t = torch.from_numpy(idx).to(dev())
    rb1 = ops.build_rulebook(t, 2, shape, 3, 2, 1, 1, 0, False, False)
    rb2 = ops.build_rulebook(t, 2, shape, 3, 2, 1, 1, 0, False, False)
    assert torch.equal(rb1.out_indices, rb2.out_indices) and torch.equal(rb1.nbr_out, rb2.nbr_out)
    o_idx, o_out, o_in, o_sh = orc.rulebook(idx, shape, 3, 2, 1, 1, orc.MODE_CONV)
    assert list(o_sh) == [21, 800, 704]
    np.testing.assert_array_equal(rb1.out_indices.cpu().numpy(), o_idx)
    np.testing.assert_array_equal(rb1.nbr_out.cpu().numpy(), o_out)
    np.testing.assert_array_equal(rb1.nbr_in.cpu().numpy(), o_in)
    rbs = ops.build_rulebook(t, 2, shape, 3, 1, 1, 1, 0, True, False)
    s_idx, s_out, s_in, _ = orc.rulebook(idx, shape, 3, 1, 0, 1, orc.MODE_SUBM)
    np.testing.assert_array_equal(rbs.nbr_out.cpu().numpy(), s_out)
    np.testing.assert_array_equal(rbs.nbr_in.cpu().numpy(), s_in)


# ------------------------------------------------------------------ conv apply
APPLY_CASES = [(4, 16), (6, 16), (16, 16), (16, 32), (34, 32), (32, 64), (64, 64), (64, 128), (256, 128), (128, 128), (32, 2), (32, 3), (2, 2), (20, 150)]


@pytest.mark.parametrize("cin,cout", APPLY_CASES)
@pytest.mark.parametrize("kind", ["subm", "conv"])
def test_conv_fwd_bwd_vs_oracle(cin, cout, kind):
    from btcdet_amd.spconv import ops
    rng = np.random.default_rng(cin * 1000 + cout)
    shape, B = (8, 20, 18), 2
    n = 700 if cin * cout <= 64 * 64 else 300
    idx = rand_indices(rng, n, B, shape)
    k, s, p = (3, 3, 3), ((1, 1, 1) if kind == "subm" else (2, 2, 2)), (1, 1, 1)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, k, s, p, (1, 1, 1), kind)
    feat = rng.standard_normal((idx.shape[0], cin)).astype(np.float32)
    W = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32) if cout % 2 == 0 else None
    dout = rng.standard_normal((o_idx.shape[0], cout)).astype(np.float32)

    f = torch.from_numpy(feat).to(dev()).requires_grad_(True)
    w = torch.from_numpy(W).to(dev()).requires_grad_(True)
    bt = None if bias is None else torch.from_numpy(bias).to(dev()).requires_grad_(True)
    out = ops.indice_conv(f, w, bt, rb)
    out.backward(torch.from_numpy(dout).to(dev()))

    ref = orc.conv_fwd(feat, W, bias, o_out)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ref)            # same fmaf chain: bit-exact
    ref_din = orc.conv_dgrad(dout, W, o_in)
    np.testing.assert_array_equal(f.grad.cpu().numpy(), ref_din)               # bit-exact
    ref_dw = orc.conv_wgrad(feat, dout, o_out, W.shape)
    # wgrad: fp32 partial sums over row splits vs a double-precision reference: rtol 1e-4 of the scale
    scale = np.abs(ref_dw).max() + 1e-6
    assert np.abs(w.grad.cpu().numpy() - ref_dw).max() <= 1e-4 * scale
    if bias is not None:
        np.testing.assert_allclose(bt.grad.cpu().numpy(), dout.sum(0), rtol=1e-4, atol=1e-4)


GLDS_SHAPES = [411, 412, 414, 418, 421, 422, 424, 221, 222, 224, 241, 242, 141, 142]


@pytest.mark.parametrize("shape_code", GLDS_SHAPES)
@pytest.mark.parametrize("kc", [16, 32, 64])
def test_conv_apply_lds_dma_instances_bit_exact(shape_code, kc):
    """every wave shape x reduction chunk of conv_apply_g (conv_apply_glds.hip), forward and dgrad, against the oracle's
    fmaf chain; the kernel is forced with btc_tune_set (kernel-selection override, results never depend on it)"""
    from btcdet_amd import _lib
    from btcdet_amd.spconv import ops
    L = _lib.lib()
    wc, ntw = (shape_code // 10) % 10, shape_code % 10
    cout = 16 * wc * ntw * (2 if wc * ntw <= 2 else 1)      # one or two column blocks
    cin = 64 if kc == 64 else (96 if kc == 32 else 48)       # 1, 3 and 3 chunks
    rng = np.random.default_rng(shape_code * 100 + kc)
    shape, B = (8, 20, 18), 2
    idx = rand_indices(rng, 500, B, shape)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), "conv")
    feat = rng.standard_normal((idx.shape[0], cin)).astype(np.float32)
    W = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    dout = rng.standard_normal((o_idx.shape[0], cout)).astype(np.float32)
    f = torch.from_numpy(feat).to(dev()).requires_grad_(True)
    w = torch.from_numpy(W).to(dev())
    bt = torch.from_numpy(bias).to(dev())
    try:
        for key, val in ((0, 2), (1, shape_code), (4, kc)):
            assert L.btc_tune_set(key, val) == 0
        out = ops.indice_conv(f, w, bt, rb)
        out.backward(torch.from_numpy(dout).to(dev()))
        torch.cuda.synchronize()
    finally:
        for key in (0, 1, 4):
            L.btc_tune_set(key, 0)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), orc.conv_fwd(feat, W, bias, o_out))
    np.testing.assert_array_equal(f.grad.cpu().numpy(), orc.conv_dgrad(dout, W, o_in))


@pytest.mark.parametrize("cin,cout", [(4, 16), (6, 16), (16, 16), (32, 2), (32, 3), (16, 32), (32, 16), (32, 32), (34, 32), (32, 64), (64, 32), (64, 64), (2, 2)])
@pytest.mark.parametrize("kind", ["subm", "conv"])
def test_wgrad_row_stationary_kernel(cin, cout, kind, exact_conv):
    """>= 4096 output rows selects conv_wgrad_rows (persistent, row-stationary); fp32 tolerance as above"""
    from btcdet_amd.spconv import ops
    rng = np.random.default_rng(cin * 100 + cout)
    shape, B = (12, 48, 44), 2
    idx = rand_indices(rng, 9000, B, shape)
    s = (1, 1, 1) if kind == "subm" else (2, 2, 2)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, (3, 3, 3), s, (1, 1, 1), (1, 1, 1), kind)
    assert o_idx.shape[0] >= 4096
    feat = rng.standard_normal((idx.shape[0], cin)).astype(np.float32)
    W = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    dout = rng.standard_normal((o_idx.shape[0], cout)).astype(np.float32)
    f = torch.from_numpy(feat).to(dev()).requires_grad_(True)
    w = torch.from_numpy(W).to(dev()).requires_grad_(True)
    out = ops.indice_conv(f, w, None, rb)
    out.backward(torch.from_numpy(dout).to(dev()))
    # large-N narrow layers take the weight-stationary conv_apply_ws kernel: same fmaf order -> still bit-exact
    np.testing.assert_array_equal(out.detach().cpu().numpy(), orc.conv_fwd(feat, W, None, o_out))
    np.testing.assert_array_equal(f.grad.cpu().numpy(), orc.conv_dgrad(dout, W, o_in))
    ref = orc.conv_wgrad(feat, dout, o_out, W.shape)
    assert np.abs(w.grad.cpu().numpy() - ref).max() <= 1e-4 * (np.abs(ref).max() + 1e-6)


@pytest.mark.parametrize("cin,cout,full_ph", [(16, 16, 4), (32, 16, 8), (16, 32, 4), (32, 32, 8), (48, 32, 4), (32, 64, 4), (64, 32, 8), (64, 64, 4)])
def test_wgrad_pipelined_kernel_both_phase_counts(cin, cout, full_ph):
    """conv_wgrad_rows_p has two instances per tile shape (PH and PH / 2 phases per offset group; the plan takes the smaller one when
    there are few row tiles): both against the oracle, and against each other up to fp32 summation order"""
    from btcdet_amd._lib import lib
    from btcdet_amd.spconv import ops
    rng = np.random.default_rng(cin * 7 + cout)
    shape, B = (12, 48, 44), 2
    idx = rand_indices(rng, 9000, B, shape)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), "subm")
    feat = rng.standard_normal((idx.shape[0], cin)).astype(np.float32)
    W = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    dout = rng.standard_normal((o_idx.shape[0], cout)).astype(np.float32)
    ref = orc.conv_wgrad(feat, dout, o_out, W.shape)
    got = []
    lib().btc_tune_set(18, 1)    # BTC_TUNE_WGRAD_X: the fp32-pipe kernel this test is about (the bf16-pipe one: tests/test_hip_wgrad_x.py)
    for ph in (0, full_ph):
        lib().btc_tune_set(5, ph)
        try:
            f = torch.from_numpy(feat).to(dev())
            w = torch.from_numpy(W).to(dev()).requires_grad_(True)
            ops.indice_conv(f, w, None, rb).backward(torch.from_numpy(dout).to(dev()))
            got.append(w.grad.cpu().numpy())
        finally:
            lib().btc_tune_set(5, 0)
            if ph == full_ph:
                lib().btc_tune_set(18, 0)
        assert np.abs(got[-1] - ref).max() <= 1e-4 * (np.abs(ref).max() + 1e-6)
    assert np.abs(got[0] - got[1]).max() <= 2e-6 * np.abs(ref).max()


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 32), (16, 32), (32, 3)])
def test_wgrad_walks_smaller_side_for_transposed_conv(cin, cout, exact_conv):
    """transposed conv with n_out > 2 n_in: btc_conv_wgrad walks nbr_in (swap path of conv_wgrad_rows)"""
    from btcdet_amd.spconv import ops
    rng = np.random.default_rng(cin + cout)
    shape, B = (6, 30, 28), 2
    idx = rand_indices(rng, 5000, B, shape)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), "transpose")
    assert o_idx.shape[0] > 2 * idx.shape[0] >= 8192
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


def test_inverse_conv_matches_oracle():
    from btcdet_amd.spconv import ops
    rng = np.random.default_rng(9)
    shape, B, cin, cout = (8, 12, 10), 2, 16, 8
    idx = rand_indices(rng, 300, B, shape)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), "conv")
    feat = rng.standard_normal((o_idx.shape[0], cin)).astype(np.float32)   # lives on the conv's OUTPUT rows
    W = rng.standard_normal((3, 3, 3, cin, cout)).astype(np.float32)
    out = ops.indice_conv(torch.from_numpy(feat).to(dev()), torch.from_numpy(W).to(dev()), None, rb, inverse=True)
    ref = orc.conv_fwd(feat, W, None, o_in)                                 # map = nbr_in (roles swapped)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


def test_maxpool_and_dense():
    from btcdet_amd.spconv import ops
    rng = np.random.default_rng(11)
    shape, B, C = (9, 14, 12), 2, 2
    idx = rand_indices(rng, 500, B, shape)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), "conv")
    feat = (rng.random((idx.shape[0], C)) - 0.2).astype(np.float32)
    feat[rng.random(feat.shape) < 0.3] = 0.5  # ties
    f = torch.from_numpy(feat).to(dev()).requires_grad_(True)
    out = ops.indice_maxpool(f, rb)
    ref = orc.maxpool_fwd(feat, o_out)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ref)
    dout = rng.standard_normal(ref.shape).astype(np.float32)
    out.backward(torch.from_numpy(dout).to(dev()))
    np.testing.assert_allclose(f.grad.cpu().numpy(), orc.maxpool_bwd(feat, ref, dout, o_in), rtol=1e-6, atol=1e-6)
    # dense fwd / bwd
    f2 = torch.from_numpy(feat).to(dev()).requires_grad_(True)
    d = ops.ToDenseFunction.apply(f2, torch.from_numpy(idx).to(dev()), B, list(shape))
    np.testing.assert_array_equal(d.detach().cpu().numpy(), orc.dense(feat, idx, B, shape))
    g = rng.standard_normal(tuple(d.shape)).astype(np.float32)
    d.backward(torch.from_numpy(g).to(dev()))
    np.testing.assert_array_equal(f2.grad.cpu().numpy(), g[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]])


def test_revoxelize_matches_oracle():
    from btcdet_amd import _lib
    rng = np.random.default_rng(13)
    B, shape, C = 2, (40, 160, 140), 6
    n = 5000
    coords = np.stack([rng.integers(0, B, n), rng.integers(0, 4, n), rng.integers(0, 30, n), rng.integers(0, 30, n)], 1).astype(np.int64)
    pts = rng.standard_normal((n, C)).astype(np.float32)
    from btcdet_amd.pass_occ_vox import revoxelize
    v, num, vc = revoxelize(torch.from_numpy(pts).to(dev()), torch.from_numpy(coords).to(dev()), B, shape)
    rv, rnum, rvc = orc.revoxelize(pts, coords)
    np.testing.assert_array_equal(vc.cpu().numpy(), rvc)
    np.testing.assert_array_equal(num.cpu().numpy(), rnum)
    np.testing.assert_array_equal(v.cpu().numpy(), rv)
    assert vc.dtype == torch.int64 and num.dtype == torch.int64


@pytest.mark.parametrize("n,c,relu", [(5000, 32, True), (130, 16, True), (70000, 64, False), (3, 128, True), (20000, 34, True)])
@pytest.mark.parametrize("training", [True, False])
def test_fused_batchnorm_relu_matches_torch(n, c, relu, training):
    """btc_bn_relu_fwd / bwd against torch's BatchNorm1d(+ReLU) in fp64: outputs, input / affine gradients, running
    statistics and num_batches_tracked (fp32 tolerance 2e-5: different reduction order)"""
    from btcdet_amd.spconv import fused_bn
    torch.manual_seed(n + c)
    x = (torch.randn(n, c, device=dev()) * 2 + 0.5)
    bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev())
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-0.1, 0.1)
        bn.running_var.uniform_(0.5, 1.5)
    ref = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev()).double()
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in bn.state_dict().items()})
    bn.train(training)
    ref.train(training)
    g = torch.randn(n, c, device=dev())
    xa = x.clone().requires_grad_(True)
    ya = fused_bn.batch_norm_relu(bn, xa, relu)
    ya.backward(g)
    xb = x.double().clone().requires_grad_(True)
    yb = ref(xb)
    if relu:
        yb = torch.relu(yb)
    yb.backward(g.double())
    tol = dict(rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(ya.detach().cpu().numpy(), yb.detach().cpu().numpy(), **tol)
    np.testing.assert_allclose(xa.grad.cpu().numpy(), xb.grad.cpu().numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(bn.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy(), rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(bn.bias.grad.cpu().numpy(), ref.bias.grad.cpu().numpy(), rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), ref.running_mean.cpu().numpy(), **tol)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), ref.running_var.cpu().numpy(), **tol)
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked)


# ------------------------------------------------------------------ the two bindings
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_compiled_binding_and_ctypes_binding_agree(dtype):
    """btcdet_amd/_btcfast (csrc/binding.cpp) and the ctypes route (_lib.py) call the same C ABI: rulebooks, the fused
    conv -> BatchNorm -> ReLU node and its gradients must be identical bit for bit"""
    from btcdet_amd import _lib, spconv
    from functools import partial
    assert _lib.fast() is not None, "the compiled binding is not built (python -c 'import __graft_entry__ as g; g.build()')"
    rng = np.random.default_rng(5)
    shape, B = (8, 24, 20), 2
    idx = rand_indices(rng, 1500, B, shape)
    idx = idx[np.lexsort((idx[:, 3], idx[:, 2], idx[:, 1], idx[:, 0]))]
    feats = rng.standard_normal((idx.shape[0], 16)).astype(np.float32)

    def run():
        torch.manual_seed(0)
        bn = partial(torch.nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        net = spconv.SparseSequential(spconv.SubMConv3d(16, 32, 3, padding=1, bias=False, indice_key="s1"), bn(32), torch.nn.ReLU(),
                                      spconv.SparseConv3d(32, 32, 3, stride=2, padding=1, bias=True, indice_key="c2"), bn(32), torch.nn.ReLU(),
                                      spconv.SparseConvTranspose3d(32, 16, 3, stride=2, padding=1, bias=False, indice_key="t3"), bn(16)).to(dev())
        f = torch.from_numpy(feats).to(dev()).to(dtype).requires_grad_(True)
        x = spconv.SparseConvTensor(f, torch.from_numpy(idx).to(dev()), list(shape), B)
        y = net(x)
        y.features.float().pow(2).sum().backward()
        torch.cuda.synchronize()
        rb = x.indice_dict["c2"]
        return [y.features.detach(), y.indices, f.grad, rb.nbr_out, rb.nbr_in, rb.out_indices] + [p.grad for p in net.parameters()] + \
               [b.clone() for b in net.buffers()]

    fast_res = run()
    saved = _lib._fast
    _lib._fast = None  # force the ctypes route
    try:
        ct_res = run()
    finally:
        _lib._fast = saved
    assert len(fast_res) == len(ct_res)
    for a, b in zip(fast_res, ct_res):
        assert a.dtype == b.dtype and torch.equal(a, b)


def test_last_arriver_reductions_stress():
    """the fence-free last-arriver protocol (csrc/btc_common.h btc_ticket_take: sc1 partials -> drained queue -> relaxed ticket; consumer:
    acquire + agent-scope loads) on grids that span all eight XCDs, every launch behind a kernel that leaves the L2s full of dirty lines,
    60 rounds with fresh data: column sums, BatchNorm statistics forward / backward (the separate-pass kernels), and the two-tensor
    sum of squares against float64 sums made by torch.  A partial read stale, or a ticket overtaking its payload, shows as a wrong sum
    (ADVICE round 5)."""
    from btcdet_amd.spconv import fused_bn
    from btcdet_amd.trainer import MeanSquare2
    g = torch.Generator(device=dev()).manual_seed(7)
    N, C = 150000, 32
    dirty = torch.empty((64 << 20) // 4, dtype=torch.float32, device=dev())
    for it in range(60):
        x = torch.randn((N, C), generator=g, device=dev()) * (1.0 + it % 5) + 0.25 * it
        dy = torch.randn((N, C), generator=g, device=dev())
        dirty.fill_(float(it))                                                    # 64 MB of dirty lines across the XCDs' L2s
        cs = fused_bn.col_sum(x)
        dirty.add_(1.0)
        y, stats = fused_bn.bn_forward(x, None, None, None, None, None, True, 0.01, 1e-3, True)
        dirty.add_(1.0)
        dx, dgamma, dbeta = fused_bn.bn_backward(x, y, dy, None, stats, True, True)
        dirty.add_(1.0)
        ms = MeanSquare2.apply(x, 1.0, dy, 3.0)
        x64, dy64 = x.double(), dy.double()
        np.testing.assert_allclose(cs.cpu().numpy(), x64.sum(0).cpu().numpy(), rtol=2e-6, atol=2e-3)
        mean, var = x64.mean(0), x64.var(0, unbiased=False)
        np.testing.assert_allclose(stats[0].cpu().numpy(), mean.cpu().numpy(), rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(stats[1].cpu().numpy(), torch.rsqrt(var + 1e-3).cpu().numpy(), rtol=2e-6)
        xh = (x64 - mean) * torch.rsqrt(var + 1e-3)
        gm = dy64 * (y > 0)               # (the mask the kernel uses: its own fp32 output -- a float64 x-hat within rounding of 0 may sit on the other side)
        np.testing.assert_allclose(dbeta.cpu().numpy(), gm.sum(0).cpu().numpy(), rtol=1e-5, atol=2e-3)
        np.testing.assert_allclose(dgamma.cpu().numpy(), (gm * xh).sum(0).cpu().numpy(), rtol=1e-5, atol=2e-3)
        ref = (x64 * x64).mean() + 3.0 * (dy64 * dy64).mean()
        assert abs(float(ms) - float(ref)) <= 2e-6 * float(ref)


def test_conv_epilogue_batch_statistics_few_offsets_few_channels():
    """a stride-2 kernel-2 layer with 4 input channels (K = 8) takes the weight-stationary kernel with only ~10 KB of panel + map LDS:
    the statistics epilogue's last arriver needs 16 + 16 B x 1024 threads of it (the launch now reserves that much) --
    mean / rstd / running statistics equal the separate statistics pass, slots left zeroed, two calls in a row agree"""
    from btcdet_amd import _lib
    from btcdet_amd.spconv import fused_bn, ops
    L = _lib.lib()
    rng = np.random.default_rng(84)
    shape, B, cin, cout = (8, 40, 48), 2, 4, 16
    idx = rand_indices(rng, 9000, B, shape)
    rb = ops.build_rulebook(torch.from_numpy(idx).to(dev()), B, shape, 2, 2, 0, 1, 0, False, False)
    m = rb.nbr_out.shape[0]
    assert rb.nbr_out.shape[1] == 8 and m >= 2048          # (>= 2048 rows: the weight-stationary kernel's range)
    feat = torch.from_numpy(rng.standard_normal((idx.shape[0], cin)).astype(np.float32)).to(dev())
    w = torch.from_numpy((rng.standard_normal((8, cin, cout)) / np.sqrt(cin)).astype(np.float32)).to(dev())
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(dev())
    beta = torch.from_numpy(rng.uniform(-0.3, 0.3, cout).astype(np.float32)).to(dev())
    outs = []
    for tune in (0, 1, 0):
        rm, rv = torch.zeros(cout, device=dev()), torch.ones(cout, device=dev())
        nbt = torch.zeros((), dtype=torch.long, device=dev())
        assert L.btc_tune_set(12, tune) == 0
        try:
            x, y, stats = fused_bn.conv_bn_forward(feat, w, None, rb.nbr_out, None, gamma, beta, rm, rv, nbt, 0.01, 1e-3, True)
            torch.cuda.synchronize()
        finally:
            L.btc_tune_set(12, 0)
        outs.append((x, y, stats, rm, rv, int(nbt)))
    assert bool((fused_bn.fuse_ws(feat.device) == 0).all())
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][0], ops.indice_conv(feat, w, None, rb))
    for a, b in ((outs[0], outs[1]), (outs[0], outs[2])):
        assert a[5] == b[5] == 1
        for i, tol in ((2, 1e-7), (3, 1e-8), (4, 1e-8)):
            np.testing.assert_allclose(a[i].cpu().numpy(), b[i].cpu().numpy(), rtol=2e-6, atol=tol)
        np.testing.assert_allclose(a[1].cpu().numpy(), b[1].cpu().numpy(), rtol=1e-5, atol=1e-5)
    ref = torch.nn.functional.batch_norm(outs[0][0], None, None, gamma, beta, True, 0.0, 1e-3).relu()
    np.testing.assert_allclose(outs[0][1].cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("cin,cout,n,dtype", [(4, 16, 6000, "f32"), (16, 16, 3000, "f32"), (64, 128, 900, "f32"), (32, 32, 120000, "f32"), (20, 150, 700, "f32"),
                                               (34, 32, 5000, "f32"), (64, 64, 5000, "bf16"), (64, 64, 12000, "bf16"), (32, 32, 30000, "bf16"),
                                               (64, 128, 9000, "bf16"), (32, 16, 5000, "bf16"), (16, 32, 5000, "bf16")])
def test_conv_epilogue_batch_statistics_match_the_separate_pass(cin, cout, n, dtype):
    """btc_conv_bn_relu_fwd gathers the BatchNorm batch statistics in the conv kernel's epilogue (csrc/bn_fuse.h) in every kernel
    family the dispatch can pick (weight-stationary, LDS-DMA, register-staged; bf16 activations): conv result bit-identical to the
    plain conv, mean / rstd / running statistics / output equal to the separate statistics pass (tuning key 12) within fp32
    rounding of fp64 sums, and the slot buffer is left zeroed (two calls in a row, different channel counts, agree)"""
    from btcdet_amd import _lib
    from btcdet_amd.spconv import fused_bn, ops
    L = _lib.lib()
    rng = np.random.default_rng(cin + cout + n)
    shape, B = ((12, 60, 90) if n > 50000 else (8, 30, 36)), 2
    idx = rand_indices(rng, n, B, shape)
    rb = ops.build_rulebook(torch.from_numpy(idx).to(dev()), B, shape, 3, 1, 1, 1, 0, True, False)
    m = rb.nbr_out.shape[0]
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    feat = torch.from_numpy(rng.standard_normal((m, cin)).astype(np.float32)).to(dev()).to(tdt)
    w = torch.from_numpy((rng.standard_normal((27, cin, cout)) / np.sqrt(cin)).astype(np.float32)).to(dev())
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(dev())
    beta = torch.from_numpy(rng.uniform(-0.3, 0.3, cout).astype(np.float32)).to(dev())
    outs = []
    for tune in (0, 1, 0):
        rm, rv = torch.zeros(cout, device=dev()), torch.ones(cout, device=dev())
        nbt = torch.zeros((), dtype=torch.long, device=dev())
        assert L.btc_tune_set(12, tune) == 0
        try:
            x, y, stats = fused_bn.conv_bn_forward(feat, w, None, rb.nbr_out, None, gamma, beta, rm, rv, nbt, 0.01, 1e-3, True)
            torch.cuda.synchronize()
        finally:
            L.btc_tune_set(12, 0)
        outs.append((x.float(), y.float(), stats, rm, rv, int(nbt)))
    plain = ops.indice_conv(feat, w, None, rb)
    assert torch.equal(outs[0][0], outs[1][0])                                   # the conv result does not depend on where the statistics are taken
    if dtype == "f32":                                                           # (bf16: the plain layer takes bf16 WEIGHT copies as well, another kernel)
        assert torch.equal(outs[0][0], plain.float())
    plain = outs[0][0]
    assert bool((fused_bn.fuse_ws(feat.device) == 0).all())                       # slots and counter left zeroed
    for a, b in ((outs[0], outs[1]), (outs[0], outs[2])):
        assert a[5] == b[5] == 1
        np.testing.assert_allclose(a[2].cpu().numpy(), b[2].cpu().numpy(), rtol=2e-6, atol=1e-7)       # mean | rstd
        np.testing.assert_allclose(a[3].cpu().numpy(), b[3].cpu().numpy(), rtol=2e-6, atol=1e-8)       # running mean
        np.testing.assert_allclose(a[4].cpu().numpy(), b[4].cpu().numpy(), rtol=2e-6, atol=1e-8)       # running var
        tol = 1e-2 if dtype == "bf16" else 1e-5
        np.testing.assert_allclose(a[1].cpu().numpy(), b[1].cpu().numpy(), rtol=tol, atol=tol)
    ref = torch.nn.functional.batch_norm(plain.float(), None, None, gamma, beta, True, 0.0, 1e-3).relu()
    np.testing.assert_allclose(outs[0][1].cpu().numpy(), ref.cpu().numpy(), rtol=(2e-2 if dtype == "bf16" else 2e-5), atol=(2e-2 if dtype == "bf16" else 2e-5))


@pytest.mark.parametrize("exact", [True, False], ids=["exact_kernels", "split_operand_kernels"])
@pytest.mark.parametrize("key,values,what", [
    (0, (1, 2), "BTC_TUNE_APPLY_KERNEL: register-staged / LDS-DMA exact kernels"),
    (2, (1, 2), "BTC_TUNE_APPLY_XCD: XCD-contiguous row-tile mapping off / on"),
    (4, (16, 32), "BTC_TUNE_APPLY_KC: reduction channels per pipeline item"),
    (13, (3, 4, 5), "BTC_TUNE_APPLY_STAGES: depth of the LDS ring"),
    (6, (128, 256, 1024), "BTC_TUNE_WGRAD_WGS: workgroups of the weight-gradient walk"),
    (11, (1,), "BTC_TUNE_WGRAD_PIPE: the two-barrier weight-gradient kernel"),
    (9, (16, 256), "BTC_TUNE_BN_FWD_KB: input per workgroup of the statistics pass"),
    (10, (32, 512), "BTC_TUNE_BN_BWD_KB: input per workgroup of the backward statistics pass")])
def test_tuning_keys_change_the_work_split_not_the_result(key, values, what, exact):
    """the tuning keys of include/btcdet_hip.h that select a tiling / a grid: a conv -> BatchNorm -> ReLU layer (64 -> 64, 9 K rows)
    forward and backward under each value against the built-in policy.  With the exact kernels (BTC_TUNE_SPLIT = 1: one fmaf chain
    per output whatever the tiling) features and input gradients are bit-identical; with the split-operand kernels and for every
    cross-row sum (weight gradient, BatchNorm statistics and parameter gradients) the grid decides the summation order: fp32
    rounding of the same sums"""
    from btcdet_amd._lib import check, lib
    from btcdet_amd import spconv
    rng = np.random.default_rng(100 + key)
    shape, B, cin, cout = (10, 44, 40), 2, 64, 64
    idx = rand_indices(rng, 9000, B, shape)
    feat = rng.standard_normal((idx.shape[0], cin)).astype(np.float32)
    dout = rng.standard_normal((idx.shape[0], cout)).astype(np.float32)
    torch.manual_seed(5)
    net = spconv.SparseSequential(spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key="s"),
                                  torch.nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01), torch.nn.ReLU()).to(dev()).train()

    def run():
        for p in net.parameters():
            p.grad = None
        f = torch.from_numpy(feat).to(dev()).requires_grad_(True)
        y = net(spconv.SparseConvTensor(f, torch.from_numpy(idx).to(dev()), list(shape), B))
        y.features.backward(torch.from_numpy(dout).to(dev()))
        return [y.features.detach().clone(), f.grad.clone()] + [p.grad.clone() for p in net.parameters()]

    def close(a, b, rel):
        return float((a - b).abs().max()) <= rel * float(a.abs().max()) + 1e-12

    check(lib().btc_tune_set(14, 1 if exact else 0), "btc_tune_set")
    try:
        ref = run()
        for v in values:
            check(lib().btc_tune_set(key, v), "btc_tune_set")
            try:
                got = run()
            finally:
                check(lib().btc_tune_set(key, 0), "btc_tune_set")
            # BatchNorm's batch statistics feed the features: bit-identical only while the statistics' own grid is untouched
            bits = exact and key not in (9, 10, 0)
            for j, (a, b) in enumerate(zip(ref, got)):
                if bits and j < 2:
                    assert torch.equal(a, b), (what, v, j)
                else:
                    assert close(a, b, 2e-5), (what, v, j)
    finally:
        check(lib().btc_tune_set(14, 0), "btc_tune_set")
