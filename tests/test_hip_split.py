"""The split-operand sparse-conv kernel (csrc/conv_apply_split.hip): fp32 in / fp32 out, every operand split exactly into three
bfloat16 pieces, the six largest piece products on the bf16 matrix pipe, one fp32 accumulator per magnitude class.  It is NOT the
bit pattern of the oracle's fmaf chain (the exact kernels stay the parity reference: `exact_conv` fixture, BTC_TUNE_SPLIT = 1), so
its bound is stated here: against the float64 product it is at least as close as the exact fp32 chain, it differs from that chain
by <= 4e-6 of the result's scale, and it is deterministic (run to run, and under any row order)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from test_hip_core import dev, rand_indices, _rb_both

pytestmark = pytest.mark.gpu


def _f64_conv(src, W, nbr, transpose):
    """float64 reference: dst[i] = sum_k src[nbr[i][k]] @ W[k] (or W[k]^T)"""
    K = nbr.shape[1]
    W = W.reshape(K, W.shape[-2], W.shape[-1]).astype(np.float64)
    s64 = src.astype(np.float64)
    out = np.zeros((nbr.shape[0], W.shape[1] if transpose else W.shape[2]))
    for k in range(K):
        rows = np.nonzero(nbr[:, k] >= 0)[0]
        if rows.size:
            out[rows] += s64[nbr[rows, k]] @ (W[k].T if transpose else W[k])
    return out


def _err(a, ref):
    d = a.astype(np.float64) - ref
    return float(np.abs(d).max() / np.abs(ref).max()), float(np.sqrt((d ** 2).mean()) / np.sqrt((ref ** 2).mean()))


@pytest.mark.parametrize("cin,cout,kind", [(64, 64, "subm"), (32, 64, "conv"), (64, 128, "subm"), (128, 128, "subm"), (256, 128, "subm"), (128, 64, "conv"),
                                           (32, 32, "big"), (64, 32, "big")])
def test_split_kernel_vs_fp64_and_exact_chain(cin, cout, kind):
    from btcdet_amd import _lib
    from btcdet_amd.spconv import ops
    L = _lib.lib()
    rng = np.random.default_rng(cin * 7 + cout)
    B = 2
    shape, n_pts = {"subm": ((12, 48, 44), 9000), "conv": ((16, 64, 64), 40000), "big": ((12, 64, 64), 26000)}[kind]   # big: > 20 K rows (32-column tiles)
    idx = rand_indices(rng, n_pts, B, shape)
    s = (2, 2, 2) if kind == "conv" else (1, 1, 1)
    kind = "subm" if kind == "big" else kind
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, (3, 3, 3), s, (1, 1, 1), (1, 1, 1), kind)
    n_out, n_in = o_out.shape[0], o_in.shape[0]
    feat = rng.standard_normal((n_in, cin)).astype(np.float32)
    W = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    dout = rng.standard_normal((n_out, cout)).astype(np.float32)
    took_fwd = L.btc_conv_split_wanted(27, cin, cout, n_out) == 1
    took_bwd = L.btc_conv_split_wanted(27, cout, cin, n_in) == 1
    assert took_fwd or took_bwd, "this case is meant to reach the split-operand kernel"

    def run():
        f = torch.from_numpy(feat).to(dev()).requires_grad_(True)
        w = torch.from_numpy(W).to(dev())
        out = ops.indice_conv(f, w, torch.from_numpy(bias).to(dev()), rb)
        out.backward(torch.from_numpy(dout).to(dev()))
        return out.detach().cpu().numpy(), f.grad.cpu().numpy()

    out, din = run()
    out2, din2 = run()
    assert np.array_equal(out, out2) and np.array_equal(din, din2)                     # deterministic
    for name, got, exact, ref64, took in (
            ("fwd", out, orc.conv_fwd(feat, W, bias, o_out), _f64_conv(feat, W, o_out, False) + bias.astype(np.float64), took_fwd),
            ("dgrad", din, orc.conv_dgrad(dout, W, o_in), _f64_conv(dout, W, o_in, True), took_bwd)):
        if not took:
            assert np.array_equal(got, exact), name                                  # the exact kernel: the oracle's bits
            continue
        assert not np.array_equal(got, exact)                                          # (it really is the other kernel)
        (mx, rms), (mx_e, rms_e) = _err(got, ref64), _err(exact, ref64)
        dmax = float(np.abs(got - exact).max() / np.abs(exact).max())
        print("%s %d -> %d %s: split vs fp64 max %.2e rms %.2e | exact chain vs fp64 max %.2e rms %.2e | split vs exact max %.2e" % (
            name, cin, cout, kind, mx, rms, mx_e, rms_e, dmax))
        assert rms <= 1.1 * rms_e and mx <= 1.5 * mx_e + 2e-7, name                    # at least as close to the true product as the fp32 chain
        assert dmax <= 4e-6, name


def test_split_kernel_row_order_and_mirror_do_not_change_bits():
    """any row permutation (btc_conv_apply_ordered's hint) and the mirrored read of a submanifold map give the same bits as the
    map order / the explicit backward map: a row's sums never depend on which rows share its tile"""
    from btcdet_amd import _lib
    from btcdet_amd._lib import check, ptr, stream_ptr
    from btcdet_amd.spconv import ops
    L = _lib.lib()
    rng = np.random.default_rng(3)
    shape, B, cin, cout = (12, 48, 44), 2, 64, 64
    idx = rand_indices(rng, 9000, B, shape)
    rb = ops.build_rulebook(torch.from_numpy(idx).to(dev()), B, shape, 3, 1, 1, 1, 0, True, False)
    n = rb.nbr_out.shape[0]
    feat = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).to(dev())
    w = torch.from_numpy((rng.standard_normal((27, cin, cout)) / 8).astype(np.float32)).to(dev())
    q = torch.empty((2, 3 * w.numel()), dtype=torch.bfloat16, device=dev())
    check(L.btc_weights_split3(ptr(w), 27, cin, cout, ptr(q[0]), ptr(q[1]), stream_ptr()), "btc_weights_split3")
    order = torch.from_numpy(rng.permutation(n).astype(np.int32)).to(dev())
    outs = []
    for o in (None, order):
        out = torch.empty((n, cout), device=dev())
        check(L.btc_conv_apply_src(0, 3, ptr(feat), n, ptr(q[1]), None, ptr(rb.nbr_out), ptr(o), n, 27, cin, cout, ptr(out), stream_ptr()), "fwd")
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    nbr_in = rb.nbr_in.contiguous()           # materialised mirror image
    dins = []
    for pass_, m in ((1, nbr_in), (2, rb.nbr_out)):
        din = torch.empty((n, cin), device=dev())
        check(L.btc_conv_apply_src(pass_, 3, ptr(outs[0]), n, ptr(q[0]), None, ptr(m), None, n, 27, cin, cout, ptr(din), stream_ptr()), "dgrad")
        dins.append(din)
    assert torch.equal(dins[0], dins[1])
    # the split kernel gathers through 32-bit byte offsets: a source whose size is unknown, or past 4 GB, is refused ON THE HOST with an
    # error code and a message -- nothing is launched, nothing traps on the device (the boundary never kills the process)
    out = torch.empty((n, cout), device=dev())
    rc = L.btc_conv_apply_ordered(0, 3, ptr(feat), ptr(q[1]), None, ptr(rb.nbr_out), None, n, 27, cin, cout, ptr(out), stream_ptr())
    assert rc != 0 and b"row count" in L.btc_last_error()
    rc = L.btc_conv_apply_src(0, 3, ptr(feat), (1 << 32) // (4 * cin) + 1, ptr(q[1]), None, ptr(rb.nbr_out), None, n, 27, cin, cout, ptr(out), stream_ptr())
    assert rc != 0 and b"4 GB" in L.btc_last_error()
    check(L.btc_conv_apply_ordered(2, 3, ptr(outs[0]), ptr(q[0]), None, ptr(rb.nbr_out), None, n, 27, cin, cout, ptr(dins[0]), stream_ptr()), "mirror")   # a submanifold source: n rows, known
    torch.cuda.synchronize()


def test_split_planes_are_an_exact_decomposition():
    """hi + mid + lo == the fp32 weight, bit for bit, and the two layouts hold the same pieces"""
    from btcdet_amd import _lib
    from btcdet_amd._lib import check, ptr, stream_ptr
    L = _lib.lib()
    rng = np.random.default_rng(0)
    K, cin, cout = 27, 32, 64
    w = (rng.standard_normal((K, cin, cout)) * np.exp(rng.uniform(-20, 20, (K, cin, cout)))).astype(np.float32)
    w[0, 0, :4] = [0.0, -0.0, 1.0, -3.0]
    wt = torch.from_numpy(w).to(dev())
    q = torch.empty((2, 3, K, cin, cout), dtype=torch.bfloat16, device=dev())
    check(L.btc_weights_split3(ptr(wt), K, cin, cout, ptr(q[0]), ptr(q[1]), stream_ptr()), "btc_weights_split3")
    planes = q[0].float().cpu().numpy().astype(np.float64)
    assert np.array_equal((planes[0] + planes[1] + planes[2]).astype(np.float32), w)
    t = q[1].view(3, K, cout, cin).float().cpu().numpy()
    assert np.array_equal(np.swapaxes(t, 2, 3), q[0].float().cpu().numpy())


def test_exact_kernel_selected_by_tune_key(exact_conv):
    from btcdet_amd import _lib
    assert _lib.lib().btc_conv_split_wanted(27, 64, 64, 20000) == 0


@pytest.mark.parametrize("cin,cout,n_pts", [(64, 64, 3500), (64, 64, 7000), (256, 128, 6000), (128, 256, 6000)])
def test_z_split_small_levels(cin, cout, n_pts):
    """levels of a few thousand rows: up to four workgroups share a tile's items, partial slabs in the stream's scratch buffer, a second
    launch adds them in order (+ bias, + the BatchNorm statistics).  Same bounds as the unsplit kernel, bit-identical run to run, and the
    fused BatchNorm statistics equal the separate pass"""
    from btcdet_amd import _lib
    from btcdet_amd.spconv import fused_bn, ops
    L = _lib.lib()
    rng = np.random.default_rng(cin + cout + n_pts)
    shape, B = (10, 40, 40), 2
    idx = rand_indices(rng, n_pts, B, shape)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), "subm")
    n = o_out.shape[0]
    assert 2500 <= n < 10000 and L.btc_conv_split_wanted(27, cin, cout, n) == 1
    feat = rng.standard_normal((n, cin)).astype(np.float32)
    W = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    f, w, b = (torch.from_numpy(a).to(dev()) for a in (feat, W, bias))
    outs = {}
    # (64 result columns from 5 K rows up: the built-in policy takes the 64 x 32 tiles, which never split -- the 64 x 64 tiles are asked for)
    assert L.btc_tune_set(1, 422 if cout == 64 else 0) == 0
    for z in (0, 1):                       # 0: the policy (z-split here), 1: never
        assert L.btc_tune_set(15, z) == 0
        try:
            outs[z] = [ops.indice_conv(f, w, b, rb).cpu().numpy() for _ in range(2)]
        finally:
            L.btc_tune_set(15, 0)
            if z == 1:
                L.btc_tune_set(1, 0)
        assert np.array_equal(outs[z][0], outs[z][1])
    assert not np.array_equal(outs[0][0], outs[1][0])          # (the split really ran: another summation order)
    ref64 = _f64_conv(feat, W, o_out, False) + bias.astype(np.float64)
    exact = orc.conv_fwd(feat, W, bias, o_out)
    (mx, rms), (mx_e, rms_e) = _err(outs[0][0], ref64), _err(exact, ref64)
    print("%d -> %d at %d rows, z-split: vs fp64 max %.2e rms %.2e | exact chain max %.2e rms %.2e" % (cin, cout, n, mx, rms, mx_e, rms_e))
    assert rms <= 1.1 * rms_e and mx <= 1.5 * mx_e + 2e-7
    # conv -> BatchNorm with the statistics taken in the reduce launch
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(dev())
    beta = torch.from_numpy(rng.uniform(-0.3, 0.3, cout).astype(np.float32)).to(dev())
    res = []
    for tune in (0, 1):
        rm, rv, nbt = torch.zeros(cout, device=dev()), torch.ones(cout, device=dev()), torch.zeros((), dtype=torch.long, device=dev())
        assert L.btc_tune_set(12, tune) == 0
        try:
            x, y, stats = fused_bn.conv_bn_forward(f, w.view(27, cin, cout), None, rb.nbr_out, None, gamma, beta, rm, rv, nbt, 0.01, 1e-3, True)
            torch.cuda.synchronize()
        finally:
            L.btc_tune_set(12, 0)
        res.append((x, y, stats, rm, rv))
    assert torch.equal(res[0][0], res[1][0])
    for a, c in zip(res[0][2:], res[1][2:]):
        np.testing.assert_allclose(a.cpu().numpy(), c.cpu().numpy(), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(res[0][1].cpu().numpy(), res[1][1].cpu().numpy(), rtol=0, atol=1e-5)
    assert bool((fused_bn.fuse_ws(f.device) == 0).all())


def test_multi_weight_split_equals_single():
    """btc_weights_split3_multi (one launch for a group's layers) writes the planes btc_weights_split3 writes, for 40 weights of mixed
    shapes (more than one table's worth)"""
    import ctypes
    from btcdet_amd import _lib
    from btcdet_amd._lib import check, ptr, stream_ptr
    L = _lib.lib()
    rng = np.random.default_rng(11)
    shapes = [(27, 32, 32), (27, 64, 64), (3, 64, 128), (27, 32, 64), (2, 128, 64), (27, 128, 32), (8, 96, 32), (1, 32, 160)] * 5
    ws = [torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(dev()) for s in shapes]
    single = []
    for w, (K, cin, cout) in zip(ws, shapes):
        q = torch.empty((2, 3 * w.numel()), dtype=torch.bfloat16, device=dev())
        check(L.btc_weights_split3(ptr(w), K, cin, cout, ptr(q[0]), ptr(q[1]), stream_ptr()), "single")
        single.append(q)
    multi = [torch.zeros_like(q) for q in single]
    n = len(ws)
    arr = lambda vals: (ctypes.c_void_p * n)(*vals)
    i32 = lambda vals: (ctypes.c_int32 * n)(*vals)
    check(L.btc_weights_split3_multi(arr([ptr(w) for w in ws]), arr([ptr(q[0]) for q in multi]), arr([ptr(q[1]) for q in multi]),
                                     i32([s[0] for s in shapes]), i32([s[1] for s in shapes]), i32([s[2] for s in shapes]), n, stream_ptr()), "multi")
    for a, b in zip(single, multi):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))


def _abs_conv(src, W, nbr, transpose):
    """S[i][c] = sum_k sum_ci |src| |W|: the scale of the terms behind every output (what rounding errors are relative to)"""
    return _f64_conv(np.abs(src), np.abs(W), nbr, transpose)


def _setup(rng, cin, cout, n_pts=9000, shape=(12, 48, 44)):
    idx = rand_indices(rng, n_pts, 2, shape)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, 2, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), "subm")
    return o_out, o_in, rb


def _run_split(feat, W, dout, rb):
    from btcdet_amd.spconv import ops
    f = torch.from_numpy(feat).to(dev()).requires_grad_(True)
    out = ops.indice_conv(f, torch.from_numpy(W).to(dev()), None, rb)
    out.backward(torch.from_numpy(dout).to(dev()))
    return out.detach().cpu().numpy(), f.grad.cpu().numpy()


@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 128)])
def test_split_kernel_under_cancellation(cin, cout):
    """unit-scale Gaussians hide what a split can lose.  Here every activation carries a large common
    offset (100 + N(0, 1): the hi piece is the same for all of them, the information sits in mid / lo) and the weights of every
    output channel sum to ~0 over the reduction, so the result is 3-4 orders of magnitude below the terms it is made of.  The
    error is measured against the scale of the TERMS (sum |a| |w|), next to the exact fp32 chain's."""
    from btcdet_amd import _lib
    rng = np.random.default_rng(cin + 1)
    o_out, o_in, rb = _setup(rng, cin, cout)
    n = o_out.shape[0]
    assert _lib.lib().btc_conv_split_wanted(27, cin, cout, n) == 1
    feat = (100.0 + rng.standard_normal((n, cin))).astype(np.float32)
    W = rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(cin)
    W = (W - W.mean(axis=(0, 1, 2, 3), keepdims=True)).astype(np.float32)          # columns sum to ~0 over (offset, cin)
    dout = (100.0 + rng.standard_normal((n, cout))).astype(np.float32)
    out, din = _run_split(feat, W, dout, rb)
    for name, got, exact, ref64, S in (("fwd", out, orc.conv_fwd(feat, W, None, o_out), _f64_conv(feat, W, o_out, False), _abs_conv(feat, W, o_out, False)),
                                       ("dgrad", din, orc.conv_dgrad(dout, W, o_in), _f64_conv(dout, W, o_in, True), _abs_conv(dout, W, o_in, True))):
        S = np.maximum(S, 1e-30)
        e_s, e_x = np.abs(got - ref64) / S, np.abs(exact - ref64) / S
        cancel = float(np.median(np.abs(ref64) / S))
        print("%s %d -> %d cancellation (|result| / sum|terms| median %.1e): split max %.2e rms %.2e | exact chain max %.2e rms %.2e" % (
            name, cin, cout, cancel, e_s.max(), np.sqrt((e_s ** 2).mean()), e_x.max(), np.sqrt((e_x ** 2).mean())))
        assert np.isfinite(got).all() and cancel < 0.05
        assert np.sqrt((e_s ** 2).mean()) <= 1.2 * np.sqrt((e_x ** 2).mean()) and e_s.max() <= 1.5 * e_x.max() + 2e-8, name
        assert e_s.max() <= 4e-7, name       # absolute statement: <= 4e-7 of the terms' scale (the fp32 chain: ~2e-7)


def test_split_kernel_exponent_range_of_activations():
    """the activation split happens in registers (conv_apply_s split2): rows scaled by 2^-100 .. 2^100, denormal and zero activations,
    values next to FLT_MAX / 2^14 -- every output is finite and as close to the float64 product as the exact chain, relative to the
    scale of its own terms"""
    from btcdet_amd import _lib
    cin = cout = 64
    rng = np.random.default_rng(77)
    o_out, o_in, rb = _setup(rng, cin, cout)
    n = o_out.shape[0]
    assert _lib.lib().btc_conv_split_wanted(27, cin, cout, n) == 1
    feat = rng.standard_normal((n, cin)) * np.exp2(rng.integers(-100, 101, size=(n, 1)).astype(np.float64))
    feat = feat.astype(np.float32)
    feat[rng.integers(0, n, 200), rng.integers(0, cin, 200)] = 0.0
    feat[rng.integers(0, n, 200), rng.integers(0, cin, 200)] = -0.0
    feat[rng.integers(0, n, 200), rng.integers(0, cin, 200)] = np.float32(1e-41)          # denormal
    feat[rng.integers(0, n, 50), rng.integers(0, cin, 50)] = np.float32(2.0 ** 113)       # large: 27 * 64 terms of it still fit fp32
    W = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    dout = (rng.standard_normal((n, cout)) * np.exp2(rng.integers(-100, 101, size=(n, 1)).astype(np.float64))).astype(np.float32)
    out, din = _run_split(feat, W, dout, rb)
    for name, got, exact, ref64, S in (("fwd", out, orc.conv_fwd(feat, W, None, o_out), _f64_conv(feat, W, o_out, False), _abs_conv(feat, W, o_out, False)),
                                       ("dgrad", din, orc.conv_dgrad(dout, W, o_in), _f64_conv(dout, W, o_in, True), _abs_conv(dout, W, o_in, True))):
        assert np.isfinite(got).all(), name
        ok = S > 1e-30                       # outputs whose every term is (sub)denormal carry no relative information
        e_s, e_x = np.abs(got - ref64)[ok] / S[ok], np.abs(exact - ref64)[ok] / S[ok]
        print("%s exponent range: %d of %d outputs compared; split max %.2e rms %.2e | exact chain max %.2e rms %.2e" % (
            name, int(ok.sum()), ok.size, e_s.max(), np.sqrt((e_s ** 2).mean()), e_x.max(), np.sqrt((e_x ** 2).mean())))
        assert e_s.max() <= 1.5 * e_x.max() + 2e-8 and np.sqrt((e_s ** 2).mean()) <= 1.2 * np.sqrt((e_x ** 2).mean()), name


@pytest.mark.parametrize("cin,cout,n_pts", [(64, 64, 3500), (64, 64, 7000), (64, 64, 16000), (256, 128, 6000), (64, 32, 26000), (32, 32, 26000)])
def test_loader_waves_give_the_same_bits(cin, cout, n_pts):
    """BTC_TUNE_SPLIT_LOADERS: 1 = the product waves issue their own LDS-DMA pieces, 2 / 4 = that many loader waves per workgroup issue them
    all (conv_apply_s, LW template parameter; 0 = the built-in policy).  Who issues a piece changes neither what lands in the LDS nor the
    order of a single product: forward and data gradient are bit-identical in every mode, with and without z-split."""
    from btcdet_amd import _lib
    from btcdet_amd.spconv import ops
    L = _lib.lib()
    rng = np.random.default_rng(cin * 7 + cout + n_pts)
    shape, B = (16, 64, 64), 2
    idx = rand_indices(rng, n_pts, B, shape)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), "subm")
    n = o_out.shape[0]
    assert L.btc_conv_split_wanted(27, cin, cout, n) == 1
    f = torch.from_numpy(rng.standard_normal((n, cin)).astype(np.float32)).to(dev()).requires_grad_(True)
    w = torch.from_numpy((rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(cin)).astype(np.float32)).to(dev())
    g = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).to(dev())
    res = []
    for mode in (1, 2, 4, 0, 1):
        assert L.btc_tune_set(17, mode) == 0
        try:
            y = ops.indice_conv(f, w, None, rb)
            (dx,) = torch.autograd.grad(y, f, g)
            res.append((y.detach().clone(), dx.clone()))
        finally:
            L.btc_tune_set(17, 0)
    for y, dx in res[1:]:
        assert torch.equal(res[0][0], y) and torch.equal(res[0][1], dx)


@pytest.mark.parametrize("cin,cout,n_pts,kind", [(32, 32, 26000, "subm"), (32, 32, 7000, "subm"), (32, 64, 16000, "subm"), (32, 32, 20000, "conv"),
                                                  (32, 64, 30000, "subm")])
def test_two_offsets_per_item_give_the_same_bits(cin, cout, n_pts, kind):
    """BTC_TUNE_SPLIT_PAIR: 32-channel reductions walk two active offsets per 64-channel item (conv_apply_s, PAIR template parameter; 1 = one
    offset per item as before, 2 = pairs wherever a tile shape has the instance, 0 = the built-in policy).  Every accumulator sees the same
    products in the same order: forward and data gradient are bit-identical -- odd and even numbers of active offsets, strided maps whose
    tiles miss most offsets, rows past the end of the last tile, with and without loader waves"""
    from btcdet_amd import _lib
    from btcdet_amd.spconv import ops
    L = _lib.lib()
    rng = np.random.default_rng(cin * 11 + cout + n_pts)
    shape, B = (16, 64, 64), 2
    idx = rand_indices(rng, n_pts, B, shape)
    stride = (2, 2, 2) if kind == "conv" else (1, 1, 1)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, (3, 3, 3), stride, (1, 1, 1), (1, 1, 1), kind)
    n_in = idx.shape[0]
    f = torch.from_numpy(rng.standard_normal((n_in, cin)).astype(np.float32)).to(dev()).requires_grad_(True)
    w = torch.from_numpy((rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(cin)).astype(np.float32)).to(dev())
    g = torch.from_numpy(rng.standard_normal((o_out.shape[0], cout)).astype(np.float32)).to(dev())
    res = []
    for pair, loaders in ((1, 0), (2, 0), (2, 1), (2, 2), (0, 0), (1, 0)):
        assert L.btc_tune_set(21, pair) == 0 and L.btc_tune_set(17, loaders) == 0
        try:
            y = ops.indice_conv(f, w, None, rb)
            (dx,) = torch.autograd.grad(y, f, g)
            res.append((y.detach().clone(), dx.clone()))
        finally:
            L.btc_tune_set(21, 0)
            L.btc_tune_set(17, 0)
    for y, dx in res[1:]:
        assert torch.equal(res[0][0], y) and torch.equal(res[0][1], dx)

