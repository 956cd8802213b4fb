"""Weight gradient on the bf16 matrix pipe (csrc/conv_wgrad_x.hip, round 5) through the C ABI.

What is asserted, per tile shape (channel pairs of the configured backbones incl. the 128 / 192 / 256-channel layers, which run as
64 x 64 blocks, and the 5 / 3-channel heads, whose missing columns are masked), submanifold / strided / transposed (the walk over the
smaller side), both phase counts (few and many row tiles):
  * fp32 activations (three exact bf16 pieces, six products): the error against a float64 product is no larger than 1.5x the fp32
    MFMA chain's (conv_wgrad_rows_p, BTC_TUNE_WGRAD_X = 1) + 2e-7 of the scale -- the tolerance north_star states for features / losses,
    made relative to what the fp32 kernel itself achieves -- and within the 1e-4 bound tests/test_hip_core.py uses for every wgrad;
  * bf16 activations: products of bf16 values are exact in fp32, so the only error is the fp32 accumulation: <= 4e-6 of the scale
    against float64 over the same bf16 inputs;
  * run-to-run bit identity; the two-call form (btc_conv_wgrad_slabs + btc_wgrad_reduce_multi, many layers in one launch) equals the
    one-call form bit for bit; an unknown row count of `feat` (n_in = -1) takes the fp32-pipe kernel (host-side refusal, no trap).
"""
import ctypes

import numpy as np
import pytest
import torch

from test_hip_core import _rb_both, dev, rand_indices

pytestmark = pytest.mark.gpu

X_KEY = 18   # BTC_TUNE_WGRAD_X


@pytest.fixture(autouse=True)
def _no_narrow_kernel():
    """the 5 / 3-channel heads of this file's strided / transposed cases would take conv_wgrad_n (tests/test_hip_wgrad_n.py) since round 6:
    BTC_TUNE_WGRAD_NARROW = 1 keeps this file on the kernel it is about"""
    from btcdet_amd._lib import lib
    assert lib().btc_tune_set(22, 1) == 0
    yield
    assert lib().btc_tune_set(22, 0) == 0


def _ref64(feat, dout, nbr_out, K, cin, cout):
    """dW[k] = sum_i feat[nbr_out[i][k]]^T dout[i] in float64 on the device"""
    f, d = feat.double(), dout.double()
    out = torch.zeros((K, cin, cout), dtype=torch.float64, device=feat.device)
    for k in range(K):
        col = nbr_out[:, k].long()
        rows = torch.nonzero(col >= 0).squeeze(1)
        if rows.numel():
            out[k] = f[col[rows]].t() @ d[rows]
    return out


def _wgrad(feat, dout, rb, cin, cout, n_in=None, slabs=False):
    """btc_conv_wgrad[_bf16] (or the two-call form) -> dW (K, cin, cout) fp32"""
    from btcdet_amd._lib import check, lib, ptr, stream_ptr
    L = lib()
    K, n_res, n_src = rb.nbr_out.shape[1], rb.nbr_out.shape[0], feat.shape[0]
    bf = feat.dtype == torch.bfloat16
    wg_bwd = None if rb.mirrored else rb.map_bwd
    n_in_arg = n_src if n_in is None else n_in
    wsb = L.btc_conv_wgrad_ws_bytes(n_res, K, cin, cout, n_in_arg if wg_bwd is not None else -1)
    wsb = max(wsb, L.btc_conv_wgrad_ws_bytes(n_res, K, cin, cout, n_src))
    ws = torch.empty((max(wsb, 256),), dtype=torch.uint8, device=feat.device)
    dw = torch.full((K, cin, cout), float("nan"), dtype=torch.float32, device=feat.device)
    pb = ptr(wg_bwd) if wg_bwd is not None else None
    if not slabs:
        fn = L.btc_conv_wgrad_bf16 if bf else L.btc_conv_wgrad
        check(fn(ptr(feat), ptr(dout), ptr(rb.nbr_out), n_res, pb, n_in_arg, K, cin, cout, ptr(dw), ptr(ws), wsb, stream_ptr()), "wgrad")
        return dw
    n = ctypes.c_int(-1)
    check(L.btc_conv_wgrad_slabs(int(bf), ptr(feat), ptr(dout), ptr(rb.nbr_out), n_res, pb, n_in_arg, None, None, K, cin, cout, ptr(dw), ptr(ws), wsb,
                                 ctypes.byref(n), stream_ptr()), "slabs")
    return dw, ws, n.value


def _case(rng, cin, cout, kind, n_vox, shape=(12, 48, 44), B=2):
    idx = rand_indices(rng, n_vox, B, shape)
    s = (1, 1, 1) if kind == "subm" else (2, 2, 2)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, (3, 3, 3), s, (1, 1, 1), (1, 1, 1), kind)
    feat = torch.from_numpy(rng.standard_normal((idx.shape[0], cin)).astype(np.float32)).to(dev())
    dout = torch.from_numpy(rng.standard_normal((o_idx.shape[0], cout)).astype(np.float32)).to(dev())
    return rb, feat, dout


SHAPES = [(32, 16), (16, 32), (32, 32), (48, 32), (32, 64), (64, 32), (64, 64), (128, 128), (256, 128), (192, 128), (64, 128), (32, 5), (64, 3)]
# (mt, nt) tile shapes per mode, larger gathered blocks first (csrc/conv_wgrad_x.hip X_BF16 / X_SPLIT)
X_TILES = {0: [(4, 4), (4, 2), (2, 4), (3, 2), (2, 2), (2, 1), (1, 2), (1, 1)], 1: [(4, 4), (4, 2), (2, 4), (3, 2), (2, 2), (2, 1), (1, 2)]}


def _x_shape(mode, cg, cc):
    """find_shape of conv_wgrad_x.hip: the gathered channels in whole blocks of 16 mt, the contiguous ones in blocks of the smallest 16 nt
    that covers them (<= 64)"""
    if cg % 16:
        return None
    nt = 1 if cc <= 16 else (2 if cc <= 32 else 4)
    while nt >= 1:
        for m, n in X_TILES[mode]:
            if n == nt and cg % (16 * m) == 0 and not (m == 3 and cg != 48):
                return m, n
        nt >>= 1
    return None


def _x_applies(mode, rb, n_src, cin, cout):
    """does the launch take conv_wgrad_x?  (the policy of sparse_conv.hip wgrad_x_wanted: the row-stationary walk, over the smaller side)"""
    n_out = rb.nbr_out.shape[0]
    swap = (not rb.mirrored) and 2 * n_src < n_out
    rows, cg, cc = (n_src, cout, cin) if swap else (n_out, cin, cout)
    return rows >= 2048 and _x_shape(mode, cg, cc) is not None


@pytest.mark.parametrize("cin,cout", SHAPES)
@pytest.mark.parametrize("kind,n_vox", [("subm", 9000), ("subm", 60000), ("conv", 30000), ("transpose", 5000)])
def test_split_wgrad_is_as_accurate_as_the_fp32_chain(cin, cout, kind, n_vox):
    from btcdet_amd._lib import check, lib
    rng = np.random.default_rng(cin * 131 + cout + n_vox)
    shape = (6, 30, 28) if kind == "transpose" else (12, 48, 44)
    rb, feat, dout = _case(rng, cin, cout, kind, n_vox, shape)
    if kind == "transpose":
        assert rb.nbr_out.shape[0] > 2 * feat.shape[0]     # the walk over the smaller (input) side
    K = rb.nbr_out.shape[1]
    ref = _ref64(feat, dout, rb.nbr_out, K, cin, cout)
    scale = float(ref.abs().max()) + 1e-12
    got = _wgrad(feat, dout, rb, cin, cout)
    again = _wgrad(feat, dout, rb, cin, cout)
    assert torch.equal(got, again), "not deterministic"
    check(lib().btc_tune_set(X_KEY, 1), "tune")
    try:
        old = _wgrad(feat, dout, rb, cin, cout)
    finally:
        check(lib().btc_tune_set(X_KEY, 0), "tune")
    assert _x_applies(1, rb, feat.shape[0], cin, cout) == (not torch.equal(got, old)), "kernel selection differs from the stated policy"
    e_new = float((got.double() - ref).abs().max()) / scale
    e_old = float((old.double() - ref).abs().max()) / scale
    r_new = float((got.double() - ref).pow(2).mean().sqrt()) / scale
    r_old = float((old.double() - ref).pow(2).mean().sqrt()) / scale
    print("%d->%d %s %d rows: max err split %.2e fp32 chain %.2e | rms %.2e / %.2e" % (cin, cout, kind, rb.nbr_out.shape[0], e_new, e_old, r_new, r_old))
    assert e_new <= 1e-4
    assert e_new <= 1.5 * e_old + 2e-7 and r_new <= 1.5 * r_old + 5e-8


@pytest.mark.parametrize("cin,cout", [(16, 16), (128, 64)] + SHAPES)
@pytest.mark.parametrize("kind,n_vox", [("subm", 9000), ("subm", 60000), ("conv", 30000), ("transpose", 5000)])
def test_bf16_wgrad_on_the_matrix_pipe(cin, cout, kind, n_vox):
    from btcdet_amd._lib import check, lib
    rng = np.random.default_rng(cin * 17 + cout * 3 + n_vox)
    shape = (6, 30, 28) if kind == "transpose" else (12, 48, 44)
    rb, feat, dout = _case(rng, cin, cout, kind, n_vox, shape)
    fb, db = feat.to(torch.bfloat16), dout.to(torch.bfloat16)
    K = rb.nbr_out.shape[1]
    ref = _ref64(fb.float(), db.float(), rb.nbr_out, K, cin, cout)
    scale = float(ref.abs().max()) + 1e-12
    got = _wgrad(fb, db, rb, cin, cout)
    assert torch.equal(got, _wgrad(fb, db, rb, cin, cout)), "not deterministic"
    check(lib().btc_tune_set(X_KEY, 1), "tune")
    try:
        old = _wgrad(fb, db, rb, cin, cout)
    finally:
        check(lib().btc_tune_set(X_KEY, 0), "tune")
    assert _x_applies(0, rb, feat.shape[0], cin, cout) == (not torch.equal(got, old)), "kernel selection differs from the stated policy"
    e_new = float((got.double() - ref).abs().max()) / scale
    e_old = float((old.double() - ref).abs().max()) / scale
    print("%d->%d %s bf16: max err %.2e (fp32-pipe kernel on widened values %.2e)" % (cin, cout, kind, e_new, e_old))
    assert e_new <= 4e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,kind,n_vox", [(64, 64, "subm", 60000), (32, 32, "subm", 9000), (64, 32, "conv", 30000), (32, 64, "transpose", 5000),
                                                 (128, 128, "subm", 9000), (48, 32, "subm", 30000), (32, 5, "subm", 60000), (16, 16, "subm", 3000), (64, 64, "subm", 2500)])
def test_one_and_two_items_in_flight_give_the_same_bits(cin, cout, kind, n_vox, dtype):
    """BTC_TUNE_WGRAD_X_DEPTH: the walk with one item of gathered rows in flight ahead of the products, and with two (a second register
    set, map rows resolved in two stages, a deeper LDS ring of map rows): same sums in the same order -> identical bits.  Every
    phase count the plans use (one phase a tile included: small launches halve PH) for both operand modes."""
    from btcdet_amd._lib import check, lib
    rng = np.random.default_rng(cin * 7 + cout + n_vox)
    shape = (6, 30, 28) if kind == "transpose" else (12, 48, 44)
    rb, feat, dout = _case(rng, cin, cout, kind, n_vox, shape)
    feat, dout = feat.to(dtype), dout.to(dtype)
    got = {}
    try:
        for depth in (1, 2):
            check(lib().btc_tune_set(20, depth), "tune")
            got[depth] = _wgrad(feat, dout, rb, cin, cout)
    finally:
        check(lib().btc_tune_set(20, 0), "tune")
    assert bool(torch.isfinite(got[1]).all()) and torch.equal(got[1], got[2])
    assert torch.equal(_wgrad(feat, dout, rb, cin, cout), got[1])      # and the built-in choice


def test_two_call_form_equals_one_call_and_batches_layers():
    from btcdet_amd._lib import check, lib, ptr, stream_ptr
    rng = np.random.default_rng(3)
    jobs, refs = [], []
    for cin, cout, kind, dt in [(64, 64, "subm", torch.float32), (32, 64, "conv", torch.float32), (32, 32, "subm", torch.bfloat16),
                                (4, 16, "subm", torch.float32), (64, 32, "transpose", torch.float32)]:
        shape = (6, 30, 28) if kind == "transpose" else (12, 48, 44)
        rb, feat, dout = _case(rng, cin, cout, kind, 5000 if kind == "transpose" else 12000, shape)
        feat, dout = feat.to(dt), dout.to(dt)
        refs.append(_wgrad(feat, dout, rb, cin, cout))
        dw, ws, n = _wgrad(feat, dout, rb, cin, cout, slabs=True)
        assert n >= 1
        jobs.append((dw, ws, n))
    P = (ctypes.c_void_p * len(jobs))(*[ptr(ws) for _, ws, _ in jobs])
    D = (ctypes.c_void_p * len(jobs))(*[ptr(dw) for dw, _, _ in jobs])
    S = (ctypes.c_int * len(jobs))(*[n for _, _, n in jobs])
    C = (ctypes.c_longlong * len(jobs))(*[dw.numel() for dw, _, _ in jobs])
    check(lib().btc_wgrad_reduce_multi(P, D, S, C, len(jobs), stream_ptr()), "reduce_multi")
    for (dw, _, _), ref in zip(jobs, refs):
        assert torch.equal(dw, ref)


def test_unknown_row_count_takes_the_fp32_pipe_kernel():
    from btcdet_amd._lib import check, lib
    rng = np.random.default_rng(5)
    rb, feat, dout = _case(rng, 64, 64, "subm", 9000)
    got = _wgrad(feat, dout, rb, 64, 64, n_in=-1)
    check(lib().btc_tune_set(X_KEY, 1), "tune")
    try:
        old = _wgrad(feat, dout, rb, 64, 64)
    finally:
        check(lib().btc_tune_set(X_KEY, 0), "tune")
    assert torch.equal(got, old)


def test_network_backward_uses_the_batched_reduction():
    """a chain of layers through the compiled binding with deferred weight gradients (the trainer's mode): gradients equal the
    per-layer reduction's bit for bit"""
    from btcdet_amd import spconv
    from btcdet_amd.spconv import ops
    if ops.fast() is None:
        pytest.skip("compiled binding not built")
    rng = np.random.default_rng(11)
    idx = rand_indices(rng, 12000, 2, (12, 48, 44))
    feat = torch.from_numpy(rng.standard_normal((idx.shape[0], 32)).astype(np.float32)).to(dev())
    torch.manual_seed(0)
    net = spconv.SparseSequential(spconv.SubMConv3d(32, 32, 3, padding=1, bias=False, indice_key="a"), torch.nn.BatchNorm1d(32), torch.nn.ReLU(),
                                  spconv.SparseConv3d(32, 64, 3, stride=2, padding=1, bias=False, indice_key="b"), torch.nn.BatchNorm1d(64), torch.nn.ReLU(),
                                  spconv.SubMConv3d(64, 64, 3, padding=1, bias=False, indice_key="c"), torch.nn.BatchNorm1d(64), torch.nn.ReLU()).to(dev()).train()
    grads = []
    for defer in (False, True):
        ops.set_defer_wgrad_join(defer)
        try:
            for p in net.parameters():
                p.grad = None
            x = spconv.SparseConvTensor(feat.clone().requires_grad_(True), torch.from_numpy(idx).to(dev()), [12, 48, 44], 2)
            net(x).features.square().sum().backward()
            ops.join_wgrad()
            torch.cuda.synchronize()
            grads.append([p.grad.clone() for p in net.parameters()])
        finally:
            ops.set_defer_wgrad_join(False)
    for a, b in zip(*grads):
        assert torch.equal(a, b)
