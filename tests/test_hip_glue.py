"""The fused helpers of csrc/glue.hip and the `_total` forms of the occupancy loss (round 5: each replaces a chain of torch elementwise /
fill / copy launches) against the torch formulations they replace -- bit-exact where the arithmetic is a copy, stated tolerance for the
fp64-accumulated sums."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,ca,cb", [(5000, 32, 2), (1, 32, 2), (777, 17, 3)])
def test_cat_features_is_cat_plus_the_zero_channels(dtype, n, ca, cb):
    from btcdet_amd.spconv import ops
    g = torch.Generator(device="cpu").manual_seed(n + ca)
    a = torch.randn((n, ca), generator=g).to(DEV).to(dtype).requires_grad_(True)
    b = torch.randn((n, cb), generator=g).to(DEV).to(dtype).requires_grad_(True)
    out = ops.cat_features(a, b)
    pad = ops._pad_amount(ca + cb, dtype) if ops.pads_in_channels(ca + cb) else 0
    ref = torch.nn.functional.pad(torch.cat((a.detach(), b.detach()), dim=1), (0, pad))
    assert out.dtype == dtype and torch.equal(out, ref)
    up = torch.randn(out.shape, generator=g).to(DEV).to(dtype)
    out.backward(up)
    assert torch.equal(a.grad, up[:, :ca]) and torch.equal(b.grad, up[:, ca:ca + cb])


def test_dense_split_equals_dense_of_the_column_slices():
    from btcdet_amd import spconv
    from btcdet_amd.spconv import ops
    rng = np.random.default_rng(4)
    shape, B, n, ca, cb = (9, 30, 40), 2, 3000, 2, 3
    lin = rng.choice(B * int(np.prod(shape)), size=n, replace=False)
    b, rem = lin // int(np.prod(shape)), lin % int(np.prod(shape))
    idx = torch.from_numpy(np.stack([b, rem // (shape[1] * shape[2]), (rem // shape[2]) % shape[1], rem % shape[2]], axis=1).astype(np.int32)).to(DEV)
    feat = torch.from_numpy(rng.standard_normal((n, ca + cb)).astype(np.float32)).to(DEV).requires_grad_(True)
    da, db = ops.dense_split(feat, idx, B, shape, ca)
    ra = spconv.SparseConvTensor(feat.detach()[:, :ca].contiguous(), idx, list(shape), B).dense()
    rb = spconv.SparseConvTensor(feat.detach()[:, ca:].contiguous(), idx, list(shape), B).dense()
    assert da.is_contiguous() and db.is_contiguous() and torch.equal(da, ra) and torch.equal(db, rb)
    ga, gb = torch.randn_like(da), torch.randn_like(db)
    (da * ga).sum().backward(retain_graph=True)       # only the first map gets a gradient: the other half must come out as zeros
    cells = (idx[:, 0].long(), idx[:, 1].long(), idx[:, 2].long(), idx[:, 3].long())
    exp_a = ga[cells[0], :, cells[1], cells[2], cells[3]]
    assert torch.equal(feat.grad[:, :ca], exp_a) and bool((feat.grad[:, ca:] == 0).all())
    feat.grad = None
    ((da * ga).sum() + (db * gb).sum()).backward()
    assert torch.equal(feat.grad, torch.cat([exp_a, gb[cells[0], :, cells[1], cells[2], cells[3]]], dim=1))


@pytest.mark.parametrize("dtype_b", [torch.float32, torch.bfloat16])
def test_stand_in_loss_matches_the_torch_formulation(dtype_b):
    from btcdet_amd.trainer import MeanSquare, MeanSquare2
    g = torch.Generator(device="cpu").manual_seed(1)
    a = torch.randn((2, 64, 50, 44), generator=g).to(DEV).requires_grad_(True)
    b = torch.randn((7001, 128), generator=g).to(DEV).to(dtype_b).requires_grad_(True)
    loss = MeanSquare2.apply(a, 1e-3, b, 2e-3)
    ref = 1e-3 * a.detach().double().pow(2).mean() + 2e-3 * b.detach().double().pow(2).mean()
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref))
    (loss * 3.0).backward()
    ka, kb = 1e-3 / a.numel(), 2e-3 / b.numel()
    exp_a = a.detach() * torch.tensor(3.0 * 2.0 * ka, device=DEV, dtype=torch.float32)
    exp_b = b.detach() * torch.tensor(3.0 * 2.0 * kb, device=DEV, dtype=torch.float32).to(dtype_b)
    assert torch.allclose(a.grad, exp_a, rtol=1e-6, atol=0) and torch.allclose(b.grad.float(), exp_b.float(), rtol=(1e-2 if dtype_b == torch.bfloat16 else 1e-6), atol=0)
    one = MeanSquare.apply(a.detach().requires_grad_(True), 1e-3)
    assert abs(float(one) - float(1e-3 * a.detach().double().pow(2).mean())) <= 2e-6 * float(one)
    # run to run identical (fixed summation order)
    assert float(MeanSquare2.apply(a, 1e-3, b, 2e-3)) == float(loss)


def test_occupancy_loss_total_form_equals_the_two_scalar_form():
    from btcdet_amd._lib import check, lib, ptr, stream_ptr
    L = lib()
    rng = np.random.default_rng(9)
    B, ncell = 2, 9 * 157 * 209
    t = lambda a: torch.from_numpy(a).to(DEV)
    logit, res, tgt = t(rng.standard_normal((B, 2, ncell)).astype(np.float32)), t(rng.standard_normal((B, 3, ncell)).astype(np.float32)), \
        t(rng.standard_normal((B, 3, ncell)).astype(np.float32))
    pos, cm, rm = t((rng.random((B, ncell)) < 0.1).astype(np.uint8)), t((rng.random((B, ncell)) < 0.4).astype(np.uint8)), t((rng.random((B, ncell)) < 0.1).astype(np.uint8))
    cw, rw = t(rng.random((B, ncell)).astype(np.float32)), t(rng.random((B, ncell)).astype(np.float32))
    ws = torch.zeros(int(L.btc_occ_loss_ws_bytes()), dtype=torch.uint8, device=DEV)
    out2, n2 = torch.empty(2, device=DEV), torch.empty(2, device=DEV)
    out3, n3 = torch.empty(3, device=DEV), torch.empty(2, device=DEV)
    args = (ptr(logit), ptr(res), ptr(tgt), ptr(pos), ptr(cm), ptr(cw), ptr(rm), ptr(rw), B, ncell, 0.11, 1.0, 0.1)
    check(L.btc_occ_loss_fwd(*args, ptr(out2), ptr(n2), ptr(ws), ws.numel(), stream_ptr()), "fwd")
    check(L.btc_occ_loss_fwd_total(*args, ptr(out3), ptr(n3), ptr(ws), ws.numel(), stream_ptr()), "fwd_total")
    assert torch.equal(out3[:2], out2) and torch.equal(n3, n2) and float(out3[2]) == float(out2[0] + out2[1])
    g = torch.tensor([0.7], device=DEV)
    dl2, dr2 = torch.zeros_like(logit), torch.zeros_like(res)
    check(L.btc_occ_loss_bwd(*args[:11], ptr(n2), ptr(g.expand(2).contiguous()), ptr(dl2), ptr(dr2), stream_ptr()), "bwd")
    dl3, dr3 = torch.full_like(logit, float("nan")), torch.full_like(res, float("nan"))     # the total form writes every cell itself
    check(L.btc_occ_loss_bwd_total(*args[:11], ptr(n3), ptr(g), ptr(dl3), ptr(dr3), stream_ptr()), "bwd_total")
    assert torch.equal(dl3, dl2) and torch.equal(dr3, dr2)
