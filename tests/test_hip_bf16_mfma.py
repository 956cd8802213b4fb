"""bfloat16 OPERANDS on the bf16 matrix pipe (csrc/conv_apply_bf16.hip; BASELINE.json configs[2] / [4]): activations AND a bf16
copy of the weights, fp32 accumulate, one rounding of the result.

Stated tolerance, per output element, against the oracle's fp32 fmaf chain over the SAME bf16-rounded operands:
    |out - ref| <= 2^-8 |ref| + 2e-6 * scale     (one bf16 ulp from the final rounding; the fp32 accumulation order of the MFMA
                                                  differs from the chain's by ~1e-7 relative of the tile's scale)
and against the fp32-WEIGHT computation (what the weight rounding costs): relative L2 error <= 4e-3 (2^-9 per weight, random)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from test_hip_core import _rb_both, dev, rand_indices

pytestmark = pytest.mark.gpu

CASES = [(32, 32, 500), (32, 64, 500), (64, 64, 500), (64, 64, 9000), (32, 64, 9000), (64, 128, 400), (128, 128, 300), (256, 128, 300),
         (128, 256, 300), (64, 16, 700), (96, 48, 500), (192, 128, 3000)]


def _check(got, ref, what):
    scale = float(np.abs(ref).max())
    err = np.abs(got - ref)
    bound = 2.0 ** -8 * np.abs(ref) + 2e-6 * scale
    worst = float((err / bound).max())
    assert worst <= 1.0, (what, worst)
    return worst


@pytest.mark.parametrize("cin,cout,n", CASES)
@pytest.mark.parametrize("kind", ["subm", "conv"])
def test_conv_bf16_operands_vs_oracle(cin, cout, n, kind):
    from btcdet_amd import _lib
    from btcdet_amd.spconv import ops
    if _lib.fast() is None:
        pytest.skip("the bf16-operand path is driven by the compiled binding")
    rng = np.random.default_rng(cin * 1000 + cout + n)
    shape, B = ((12, 48, 44), 2) if n > 2500 else ((8, 20, 18), 2)
    idx = rand_indices(rng, n, B, shape)
    s = (1, 1, 1) if kind == "subm" else (2, 2, 2)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, (3, 3, 3), s, (1, 1, 1), (1, 1, 1), kind)
    feat = orc.bf16_round(rng.standard_normal((idx.shape[0], cin)).astype(np.float32))
    W = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    Wq = orc.bf16_round(W)
    bias = rng.standard_normal(cout).astype(np.float32)
    dout = orc.bf16_round(rng.standard_normal((o_idx.shape[0], cout)).astype(np.float32))
    f = torch.from_numpy(feat).to(dev()).to(torch.bfloat16).requires_grad_(True)
    w = torch.from_numpy(W).to(dev()).requires_grad_(True)
    bt = torch.from_numpy(bias).to(dev()).requires_grad_(True)
    assert _lib.lib().btc_conv_bf16w_supported(27, cin, cout) == 1
    out = ops.indice_conv(f, w, bt, rb)
    assert out.dtype == torch.bfloat16
    out.backward(torch.from_numpy(dout).to(dev()).to(torch.bfloat16))
    got = out.detach().float().cpu().numpy()
    ref = orc.conv_fwd(feat, Wq, bias, o_out)
    a = _check(got, ref, "forward")
    ref32 = orc.conv_fwd(feat, W, bias, o_out)
    rel = float(np.linalg.norm(got - ref32) / np.linalg.norm(ref32))
    assert rel <= 4e-3 + 2.0 ** -8, rel           # weight rounding + the result's own bf16 rounding
    if _lib.lib().btc_conv_bf16w_supported(27, cout, cin) == 1:     # dgrad reduces over Cout
        gd = f.grad.float().cpu().numpy()
        _check(gd, orc.conv_dgrad(dout, Wq, o_in), "dgrad")
    # weight gradient: fp32 accumulation from the bf16 activations (unchanged kernel)
    ref_dw = orc.conv_wgrad(feat, dout, o_out, W.shape)
    assert np.abs(w.grad.cpu().numpy() - ref_dw).max() <= 1e-4 * (np.abs(ref_dw).max() + 1e-6)
    # a second forward after an in-place weight update must see the new weights (the bf16 copy is keyed by the version counter)
    with torch.no_grad():
        w.mul_(0.5)
    out2 = ops.indice_conv(f.detach(), w.detach(), None, rb).float().cpu().numpy()
    _check(out2, orc.conv_fwd(feat, orc.bf16_round(0.5 * W), None, o_out), "forward after update")


def test_weights_to_bf16_layouts():
    from btcdet_amd._lib import check, lib, ptr, stream_ptr
    rng = np.random.default_rng(3)
    K, ci, co = 27, 48, 80
    W = rng.standard_normal((K, ci, co)).astype(np.float32)
    w = torch.from_numpy(W).to(dev())
    q = torch.empty((2, W.size), dtype=torch.bfloat16, device=dev())
    check(lib().btc_weights_to_bf16(ptr(w), K, ci, co, ptr(q[0]), ptr(q[1]), stream_ptr()), "btc_weights_to_bf16")
    a = q[0].float().cpu().numpy().reshape(K, ci, co)
    b = q[1].float().cpu().numpy().reshape(K, co, ci)
    np.testing.assert_array_equal(a, orc.bf16_round(W))
    np.testing.assert_array_equal(b, orc.bf16_round(W).transpose(0, 2, 1))


@pytest.mark.parametrize("cin,cout,n,kind", [(32, 32, 26000, "subm"), (32, 64, 9000, "subm"), (32, 32, 20000, "conv"), (32, 16, 12000, "subm")])
def test_bf16_two_offsets_per_item_give_the_same_bits(cin, cout, n, kind):
    """conv_apply_b's PAIR instances (two active offsets per 64-channel item on 32-channel reductions, BTC_TUNE_SPLIT_PAIR as for the
    split-operand kernel): forward and data gradient bit-identical to one offset per item"""
    from btcdet_amd import _lib
    from btcdet_amd.spconv import ops
    if _lib.fast() is None:
        pytest.skip("the bf16-operand path is driven by the compiled binding")
    L = _lib.lib()
    rng = np.random.default_rng(cin * 13 + cout + n)
    shape, B = (16, 64, 64), 2
    idx = rand_indices(rng, n, B, shape)
    s = (1, 1, 1) if kind == "subm" else (2, 2, 2)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, (3, 3, 3), s, (1, 1, 1), (1, 1, 1), kind)
    f = torch.from_numpy(rng.standard_normal((idx.shape[0], cin)).astype(np.float32)).to(dev()).to(torch.bfloat16).requires_grad_(True)
    w = torch.from_numpy((rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(cin)).astype(np.float32)).to(dev())
    g = torch.from_numpy(rng.standard_normal((o_out.shape[0], cout)).astype(np.float32)).to(dev()).to(torch.bfloat16)
    res = []
    for pair in (1, 2, 0, 1):
        assert L.btc_tune_set(21, pair) == 0
        try:
            y = ops.indice_conv(f, w, None, rb)
            (dx,) = torch.autograd.grad(y, f, g)
            res.append((y.detach().clone(), dx.clone()))
        finally:
            L.btc_tune_set(21, 0)
    assert res[0][0].dtype == torch.bfloat16
    for y, dx in res[1:]:
        assert torch.equal(res[0][0], y) and torch.equal(res[0][1], dx)
