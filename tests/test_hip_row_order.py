"""Row-order hints (csrc/row_order.hip): btc_row_orders against a numpy stable sort by first present offset in 2048-row blocks (exact), and the
ordered apply / weight-gradient entry points against the plain ones -- forward and dgrad bit-identical for ANY permutation
(a row's sum only involves its own map row), the weight gradient equal up to fp32 summation order and run-to-run identical."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from btcdet_amd import _lib
    return _lib


def _maps(seed, n_in, shape, stride):
    """a strided 3x3x3 rulebook and a SubM rulebook on random coordinates"""
    from btcdet_amd.spconv import ops
    rng = np.random.default_rng(seed)
    cells = rng.choice(shape[0] * shape[1] * shape[2] * 2, size=n_in, replace=False)
    b, rem = np.divmod(np.sort(cells), shape[0] * shape[1] * shape[2])
    z, rem = np.divmod(rem, shape[1] * shape[2])
    y, x = np.divmod(rem, shape[2])
    idx = torch.from_numpy(np.stack([b, z, y, x], 1).astype(np.int32)).cuda()
    rb_s = ops.build_rulebook(idx, 2, shape, 3, stride, 1, 1, 0, False, False)
    rb_m = ops.build_rulebook(idx, 2, shape, 3, 1, 1, 1, 0, True, False)
    return rb_s, rb_m


def _expected_order(nbr):
    nbr = nbr.cpu().numpy()
    has = nbr >= 0
    key = np.where(has.any(1), has.argmax(1), nbr.shape[1])   # first present offset, K for a row without neighbours
    order = np.arange(nbr.shape[0], dtype=np.int32)
    for s in range(0, nbr.shape[0], 2048):    # stable sort by that key inside blocks of 2048 consecutive rows
        order[s:s + 2048] = s + np.argsort(key[s:s + 2048], kind="stable")
    return order


def test_row_orders_is_the_blockwise_stable_sort_by_first_offset():
    from btcdet_amd.spconv import ops
    rb_s, rb_m = _maps(0, 9000, (21, 40, 44), 2)
    assert rb_s.order_out is not None and rb_s.order_in is not None and rb_m.order_out is None
    maps = [rb_s.nbr_out, rb_s.nbr_in, rb_m.nbr_out, rb_s.nbr_out[:0], rb_m.nbr_in[:777]]
    got = ops.row_orders(maps)
    for m, o in zip(maps, got):
        np.testing.assert_array_equal(o.cpu().numpy(), _expected_order(m))
    np.testing.assert_array_equal(rb_s.order_out.cpu().numpy(), _expected_order(rb_s.nbr_out))
    np.testing.assert_array_equal(rb_s.order_in.cpu().numpy(), _expected_order(rb_s.nbr_in))
    # more maps than one job table carries
    many = [rb_m.nbr_out[i * 100:(i + 1) * 100 + 50] for i in range(70)]
    for m, o in zip(many, ops.row_orders(many)):
        np.testing.assert_array_equal(o.cpu().numpy(), _expected_order(m))


@pytest.mark.parametrize("operands", [0, 1, 2])
@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 64), (16, 32), (128, 128), (6, 16)])
def test_ordered_apply_is_bit_identical(operands, cin, cout):
    L = _lib()
    lib, ptr, check, sp = L.lib(), L.ptr, L.check, L.stream_ptr
    if operands and (cin % 16 or cout % 16):
        pytest.skip("bf16 activations need channel counts that are multiples of 16")
    rb_s, rb_m = _maps(1, 7001, (17, 36, 40), 2)
    torch.manual_seed(0)
    dt = torch.bfloat16 if operands else torch.float32
    for rb in (rb_s, rb_m):
        K = rb.K
        w = (torch.randn((K, cin, cout), device="cuda") * 0.1)
        for pass_, nbr, n_src in ((0, rb.nbr_out, rb.n_in), (1, rb.nbr_in, rb.n_out)):
            cred, cres = (cin, cout) if pass_ == 0 else (cout, cin)
            if operands == 2 and not lib.btc_conv_bf16w_supported(K, cred, cres):
                continue
            n = nbr.shape[0]
            src = torch.randn((n_src, cred), device="cuda").to(dt)
            wq = w
            if operands == 2:
                q = torch.empty((2, w.numel()), dtype=torch.bfloat16, device="cuda")
                check(lib.btc_weights_to_bf16(ptr(w), K, cin, cout, ptr(q[0]), ptr(q[1]), sp()), "w2bf")
                wq = q[1] if pass_ == 0 else q[0]
            bias = torch.randn((cout,), device="cuda") if pass_ == 0 else None
            outs = []
            orders = [None, ops_order(nbr), torch.randperm(n, device="cuda").int(), torch.arange(n - 1, -1, -1, device="cuda").int()]
            for o in orders:
                dst = torch.full((n, cres), float("nan"), device="cuda").to(dt)
                check(lib.btc_conv_apply_ordered(pass_, operands, ptr(src), ptr(wq), ptr(bias), ptr(nbr), ptr(o), n, K, cin, cout, ptr(dst), sp()),
                      "btc_conv_apply_ordered")
                outs.append(dst.float().cpu().numpy())
            assert np.isfinite(outs[0]).all()
            for o in outs[1:]:
                np.testing.assert_array_equal(o, outs[0])


def ops_order(nbr):
    from btcdet_amd.spconv import ops
    return ops.row_orders([nbr])[0]


@pytest.mark.parametrize("bf", [0, 1])
def test_ordered_wgrad(bf):
    L = _lib()
    lib, ptr, check, sp = L.lib(), L.ptr, L.check, L.stream_ptr
    rb_s, rb_m = _maps(2, 12000, (21, 40, 44), 2)
    torch.manual_seed(1)
    cin, cout = 32, 64
    dt = torch.bfloat16 if bf else torch.float32
    for rb in (rb_s, rb_m):
        K = rb.K
        feat = torch.randn((rb.n_in, cin), device="cuda").to(dt)
        dout = torch.randn((rb.n_out, cout), device="cuda").to(dt)
        wsb = lib.btc_conv_wgrad_ws_bytes(rb.n_out, K, cin, cout, rb.n_in)
        ws = torch.empty((wsb,), dtype=torch.uint8, device="cuda")
        o_out, o_in = ops_order(rb.nbr_out), ops_order(rb.nbr_in)
        res = []
        for oo, oi in ((None, None), (o_out, o_in), (o_out, o_in)):
            dw = torch.full((K, cin, cout), float("nan"), device="cuda")
            check(lib.btc_conv_wgrad_ordered(bf, ptr(feat), ptr(dout), ptr(rb.nbr_out), rb.n_out, ptr(rb.nbr_in), rb.n_in, ptr(oo), ptr(oi), K, cin, cout,
                                             ptr(dw), ptr(ws), wsb, sp()), "btc_conv_wgrad_ordered")
            res.append(dw.cpu().numpy())
        np.testing.assert_array_equal(res[1], res[2])                     # deterministic for a given order
        # the bf16-pipe walk with one / two items of gathered rows in flight (BTC_TUNE_WGRAD_X_DEPTH): its two-stage resolution of the
        # rows through `order` must give the same bits as the direct one
        try:
            for depth in (1, 2):
                check(lib.btc_tune_set(20, depth), "tune")
                dw = torch.full((K, cin, cout), float("nan"), device="cuda")
                check(lib.btc_conv_wgrad_ordered(bf, ptr(feat), ptr(dout), ptr(rb.nbr_out), rb.n_out, ptr(rb.nbr_in), rb.n_in, ptr(o_out), ptr(o_in), K,
                                                 cin, cout, ptr(dw), ptr(ws), wsb, sp()), "btc_conv_wgrad_ordered")
                np.testing.assert_array_equal(dw.cpu().numpy(), res[1])
        finally:
            check(lib.btc_tune_set(20, 0), "tune")
        scale = np.abs(res[0]).max()
        assert np.abs(res[1] - res[0]).max() <= 2e-6 * scale * np.sqrt(rb.n_out / 64.0) + 1e-30   # fp32 summation order only


def test_module_results_do_not_depend_on_the_hint(monkeypatch):
    """SparseConv3d / SparseInverseConv3d forward + backward with and without the hints: same bits (wgrad: same values up to
    summation order -- it does not use the hints)"""
    from btcdet_amd import spconv
    from btcdet_amd.spconv import ops
    rb_s, _ = _maps(3, 6000, (17, 36, 40), 2)
    idx = rb_s.in_indices
    torch.manual_seed(2)
    down = spconv.SparseConv3d(16, 32, 3, stride=2, padding=1, bias=False, indice_key="d").cuda()
    up = spconv.SparseInverseConv3d(32, 16, 3, indice_key="d", bias=False).cuda()
    feats = torch.randn((idx.shape[0], 16), device="cuda")
    outs = []
    for on in (True, False):
        monkeypatch.setattr(ops, "ROW_ORDER", on)
        f = feats.clone().requires_grad_(True)
        x = spconv.SparseConvTensor(f, idx, (17, 36, 40), 2)
        y = up(down(x))
        rb = x.indice_dict["d"] if "d" in x.indice_dict else None
        assert rb is None or (rb.order_out is not None) == on
        y.features.square().sum().backward()
        outs.append((y.features.detach().cpu().numpy(), f.grad.cpu().numpy()))
        down.weight.grad = up.weight.grad = None
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
