"""Weight gradient of a layer with a narrow result side (csrc/conv_wgrad_n.hip, round 6) through the C ABI.

The kernel walks the layer's INPUT rows (x read once, dy gathered through the backward map -- for a submanifold layer the forward map
read mirrored, requested by passing nbr_in == nbr_out) with the K offsets and the <= 8 result channels as one matrix dimension on the fp32
matrix pipe; a layer with a narrow INPUT (the 4- / 6-channel first layers) is walked the other way round: over its output rows, the features
gathered through nbr_out.  Asserted, for the 5 / 3-channel heads and other narrow shapes, submanifold / strided / transposed, row counts that are not
multiples of 4, both activation types:
  * fp32: the error against a float64 product is no larger than 1.5x the fp32 MFMA chain's (conv_wgrad_rows_p: BTC_TUNE_WGRAD_NARROW = 1
    and BTC_TUNE_WGRAD_X = 1) + 2e-7 of the scale -- or 1e-6 of it: the strided cases sum over the other side of the rulebook, in other
    partial sums -- and within the 1e-4 bound tests/test_hip_core.py uses for every wgrad;
  * bf16 activations: <= 4e-6 of the scale against float64 over the same bf16 inputs (exact products, fp32 accumulation);
  * run-to-run bit identity, any number of workgroups (BTC_TUNE_WGRAD_WGS); the two-call form (btc_conv_wgrad_slabs +
    btc_wgrad_reduce_multi) equals the one-call form bit for bit;
  * the policy: taken only with a backward map (or the mirror request), >= 2048 input rows, <= 8 result channels, K * Cout <= 144, input
    channels a multiple of 16; everything else runs the kernels it ran before (same bits with the switch off).
"""
import ctypes

import numpy as np
import pytest
import torch

from test_hip_core import _rb_both, dev, rand_indices
from test_hip_wgrad_x import _ref64

pytestmark = pytest.mark.gpu

N_KEY, X_KEY, WGS_KEY = 22, 18, 6   # BTC_TUNE_WGRAD_NARROW, BTC_TUNE_WGRAD_X, BTC_TUNE_WGRAD_WGS


def _wgrad(feat, dout, rb, cin, cout, slabs=False, legacy=False):
    """btc_conv_wgrad[_bf16] (or the two-call form) -> dW (K, cin, cout) fp32.  A submanifold rulebook's single map goes in twice (the
    mirror request) unless `legacy` (NULL, as callers before round 6 passed it)"""
    from btcdet_amd._lib import check, lib, ptr, stream_ptr
    L = lib()
    K, n_res, n_src = rb.nbr_out.shape[1], rb.nbr_out.shape[0], feat.shape[0]
    bf = feat.dtype == torch.bfloat16
    pb = None if (rb.mirrored and legacy) else ptr(rb.map_bwd)
    wsb = L.btc_conv_wgrad_ws_bytes(n_res, K, cin, cout, n_src)
    ws = torch.empty((max(wsb, 256),), dtype=torch.uint8, device=feat.device)
    dw = torch.full((K, cin, cout), float("nan"), dtype=torch.float32, device=feat.device)
    if not slabs:
        fn = L.btc_conv_wgrad_bf16 if bf else L.btc_conv_wgrad
        check(fn(ptr(feat), ptr(dout), ptr(rb.nbr_out), n_res, pb, n_src, K, cin, cout, ptr(dw), ptr(ws), wsb, stream_ptr()), "wgrad")
        return dw
    n = ctypes.c_int(-1)
    check(L.btc_conv_wgrad_slabs(int(bf), ptr(feat), ptr(dout), ptr(rb.nbr_out), n_res, pb, n_src, None, None, K, cin, cout, ptr(dw), ptr(ws), wsb,
                                 ctypes.byref(n), stream_ptr()), "slabs")
    return dw, ws, n.value


def _case(rng, cin, cout, kind, n_vox, k=(3, 3, 3)):
    shape, B = ((6, 30, 28) if kind == "transpose" else (12, 48, 44)), 2
    idx = rand_indices(rng, n_vox, B, shape)
    s = (1, 1, 1) if kind == "subm" else (2, 2, 2)
    p = tuple(kk // 2 for kk in k) if kind == "subm" else ((1, 1, 1) if k == (3, 3, 3) else (0, 0, 0))
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, k, s, p, (1, 1, 1), kind)
    feat = torch.from_numpy(rng.standard_normal((idx.shape[0], cin)).astype(np.float32)).to(dev())
    dout = torch.from_numpy(rng.standard_normal((o_idx.shape[0], cout)).astype(np.float32)).to(dev())
    return rb, feat, dout


def _tuned(pairs, fn):
    from btcdet_amd._lib import check, lib
    try:
        for k, v in pairs:
            check(lib().btc_tune_set(k, v), "tune")
        return fn()
    finally:
        for k, _ in pairs:
            check(lib().btc_tune_set(k, 0), "tune")


def _applies(rb, n_src, cin, cout):
    """btc_wgrad_n_kind + the row counts of sparse_conv.hip: narrow result (walk over the input rows) or narrow input (over the output rows)"""
    K, n_out = rb.nbr_out.shape[1], rb.nbr_out.shape[0]
    if cout <= 8 and K * cout <= 176 and cin % 16 == 0:
        return n_src >= 2048
    return cin <= 8 and K * cin <= 176 and cout % 16 == 0 and n_out >= 2048 and (rb.mirrored or 2 * n_src >= n_out)


# narrow result: the 5 / 3-channel heads ...; narrow input: the 4- / 6-channel first layers (27 x 6 = 162 columns: the 11-tile instances)
SHAPES = [(32, 5), (64, 3), (16, 5), (48, 2), (32, 1), (96, 4), (64, 5), (32, 6), (6, 16), (4, 16), (4, 32), (3, 48), (6, 64)]
CASES = [("subm", 9001), ("subm", 60003), ("conv", 30002), ("transpose", 5001)]


@pytest.mark.parametrize("cin,cout", SHAPES)
@pytest.mark.parametrize("kind,n_vox", CASES)
def test_narrow_wgrad_is_as_accurate_as_the_fp32_chain(cin, cout, kind, n_vox):
    rng = np.random.default_rng(cin * 131 + cout + n_vox)
    rb, feat, dout = _case(rng, cin, cout, kind, n_vox)
    K = rb.nbr_out.shape[1]
    got = _wgrad(feat, dout, rb, cin, cout)
    if not _applies(rb, feat.shape[0], cin, cout):      # (a narrow input whose rulebook has the smaller side there: the kernels of before)
        assert kind == "transpose" and cin <= 8
        assert torch.equal(got, _tuned([(N_KEY, 1)], lambda: _wgrad(feat, dout, rb, cin, cout)))
        return
    ref = _ref64(feat, dout, rb.nbr_out, K, cin, cout)
    scale = float(ref.abs().max()) + 1e-12
    assert bool(torch.isfinite(got).all())
    assert torch.equal(got, _wgrad(feat, dout, rb, cin, cout)), "not deterministic"
    old = _tuned([(N_KEY, 1), (X_KEY, 1)], lambda: _wgrad(feat, dout, rb, cin, cout))
    assert not torch.equal(got, old), "the narrow kernel was not taken"
    e_new, e_old = float((got.double() - ref).abs().max()) / scale, float((old.double() - ref).abs().max()) / scale
    r_new, r_old = float((got.double() - ref).pow(2).mean().sqrt()) / scale, float((old.double() - ref).pow(2).mean().sqrt()) / scale
    print("%d->%d %s %d input rows: max err narrow %.2e fp32 chain %.2e | rms %.2e / %.2e" % (cin, cout, kind, feat.shape[0], e_new, e_old, r_new, r_old))
    assert e_new <= 1e-4
    # (the same arithmetic in another order -- the walk is over the other side of the rulebook: fp32 rounding noise either way)
    assert e_new <= max(1.5 * e_old + 2e-7, 1e-6) and r_new <= max(1.5 * r_old + 5e-8, 2e-7)
    # any number of workgroups (one workgroup: four fp32 chains over a quarter of the rows each -- the general bound); the two-call
    # form: the same bits as the one-call form
    for wgs in (1, 7, 1024):
        g2 = _tuned([(WGS_KEY, wgs)], lambda: _wgrad(feat, dout, rb, cin, cout))
        assert float((g2.double() - ref).abs().max()) / scale <= (1e-4 if wgs < 64 else max(1.5 * e_old + 2e-7, 1e-6))
    dw, ws, n = _wgrad(feat, dout, rb, cin, cout, slabs=True)
    assert n >= 1
    from btcdet_amd._lib import check, lib, ptr, stream_ptr
    P, D = (ctypes.c_void_p * 1)(ptr(ws)), (ctypes.c_void_p * 1)(ptr(dw))
    S, C = (ctypes.c_int * 1)(n), (ctypes.c_longlong * 1)(dw.numel())
    check(lib().btc_wgrad_reduce_multi(P, D, S, C, 1, stream_ptr()), "reduce_multi")
    assert torch.equal(dw, got)


@pytest.mark.parametrize("cin,cout", SHAPES)
@pytest.mark.parametrize("kind,n_vox", CASES)
def test_narrow_wgrad_bf16_activations(cin, cout, kind, n_vox):
    rng = np.random.default_rng(cin * 17 + cout * 3 + n_vox)
    rb, feat, dout = _case(rng, cin, cout, kind, n_vox)
    fb, db = feat.to(torch.bfloat16), dout.to(torch.bfloat16)
    K = rb.nbr_out.shape[1]
    ref = _ref64(fb.float(), db.float(), rb.nbr_out, K, cin, cout)
    scale = float(ref.abs().max()) + 1e-12
    got = _wgrad(fb, db, rb, cin, cout)
    assert torch.equal(got, _wgrad(fb, db, rb, cin, cout)), "not deterministic"
    old = _tuned([(N_KEY, 1)], lambda: _wgrad(fb, db, rb, cin, cout))
    assert _applies(rb, feat.shape[0], cin, cout) == (not torch.equal(got, old)), "kernel selection differs from the stated policy"
    e_new = float((got.double() - ref).abs().max()) / scale
    print("%d->%d %s bf16: max err %.2e" % (cin, cout, kind, e_new))
    assert e_new <= 4e-6


def test_kernel_2_stride_2_layer_with_8_result_channels():
    """K = 8 offsets x 8 channels = 64 columns: four of the nine column tiles, the rest walk as column 0 and are not written"""
    rng = np.random.default_rng(11)
    rb, feat, dout = _case(rng, 32, 8, "conv", 40001, k=(2, 2, 2))
    K = rb.nbr_out.shape[1]
    assert K == 8
    ref = _ref64(feat, dout, rb.nbr_out, K, 32, 8)
    scale = float(ref.abs().max()) + 1e-12
    got = _wgrad(feat, dout, rb, 32, 8)
    old = _tuned([(N_KEY, 1), (X_KEY, 1)], lambda: _wgrad(feat, dout, rb, 32, 8))
    assert not torch.equal(got, old)
    assert float((got.double() - ref).abs().max()) / scale <= 1.5 * float((old.double() - ref).abs().max()) / scale + 2e-7


@pytest.mark.parametrize("cin,cout,kind,n_vox,why", [(32, 5, "subm", 1500, "fewer than 2048 input rows"), (32, 16, "subm", 9000, "16 result channels"),
                                                      (20, 5, "subm", 9000, "input channels not a multiple of 16"), (32, 7, "subm", 9000, "27 x 7 = 189 columns"),
                                                      (8, 16, "subm", 9000, "27 x 8 = 216 columns"), (6, 24, "subm", 9000, "result channels not a multiple of 16"),
                                                      (6, 16, "subm", 1500, "fewer than 2048 output rows")])
def test_everything_else_runs_what_it_ran_before(cin, cout, kind, n_vox, why):
    rng = np.random.default_rng(cin + cout + n_vox)
    rb, feat, dout = _case(rng, cin, cout, kind, n_vox)
    got = _wgrad(feat, dout, rb, cin, cout)
    off = _tuned([(N_KEY, 1)], lambda: _wgrad(feat, dout, rb, cin, cout))
    assert torch.equal(got, off), why


def test_a_125_offset_kernel_stays_on_the_kernels_of_before():
    """K <= 64 (a row past the end is addressed as N_RECORDS + 4 k': more offsets would wrap the 32-bit offset)"""
    rng = np.random.default_rng(9)
    rb, feat, dout = _case(rng, 16, 1, "subm", 6000, k=(5, 5, 5))
    assert rb.nbr_out.shape[1] == 125
    got = _wgrad(feat, dout, rb, 16, 1)
    assert torch.equal(got, _tuned([(N_KEY, 1)], lambda: _wgrad(feat, dout, rb, 16, 1)))
    ref = _ref64(feat, dout, rb.nbr_out, 125, 16, 1)
    assert float((got.double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


def test_without_a_backward_map_the_output_rows_are_walked():
    """a caller that passes NULL for a submanifold layer's backward map (every caller before round 6) gets the kernels of before"""
    rng = np.random.default_rng(2)
    rb, feat, dout = _case(rng, 32, 5, "subm", 20000)
    legacy = _wgrad(feat, dout, rb, 32, 5, legacy=True)
    off = _tuned([(N_KEY, 1)], lambda: _wgrad(feat, dout, rb, 32, 5))
    assert torch.equal(legacy, off)
    ref = _ref64(feat, dout, rb.nbr_out, 27, 32, 5)
    assert float((_wgrad(feat, dout, rb, 32, 5).double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


def test_a_gradient_row_of_infinities_stays_where_it_belongs():
    """rows past the end and absent neighbours are out-of-range loads, never a multiplication by a present row's value: dW[k] is
    infinite exactly where the float64 sum over the PRESENT pairs is (the kernels that walk the output rows multiply an absent
    neighbour's zeros by the infinity: NaN in every offset)"""
    rng = np.random.default_rng(4)
    rb, feat, dout = _case(rng, 32, 5, "subm", 9001)
    dout[0] = float("inf")
    ref = _ref64(feat, dout, rb.nbr_out, 27, 32, 5)
    got = _wgrad(feat, dout, rb, 32, 5)
    fin = torch.isfinite(ref)
    assert 0 < int((~fin).sum()) < fin.numel()
    assert torch.equal(torch.isfinite(got), fin) and torch.equal(got[~fin].double(), ref[~fin])


def test_autograd_path_takes_it():
    """ops.indice_conv's backward hands the single map of a submanifold rulebook in twice: dW of the 32 -> 5 head equals the C-ABI call"""
    from btcdet_amd.spconv import ops
    rng = np.random.default_rng(6)
    rb, feat, dout = _case(rng, 32, 5, "subm", 30000)
    f = feat.clone().requires_grad_(True)
    w = torch.from_numpy((rng.standard_normal((3, 3, 3, 32, 5)) / 6).astype(np.float32)).to(dev()).requires_grad_(True)
    y = ops.indice_conv(f, w, None, rb)
    dx, dw = torch.autograd.grad(y, (f, w), dout)
    ops.join_wgrad()
    torch.cuda.synchronize()
    assert torch.equal(dw.reshape(27, 32, 5), _wgrad(feat, dout, rb, 32, 5))
