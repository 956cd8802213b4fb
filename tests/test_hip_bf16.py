"""BASELINE.json configs[2] ("bf16 features"): activations are bfloat16 in HBM, weights / accumulation / statistics fp32.
The conv forward and dgrad results must be the bf16 rounding (ties to even) of exactly the oracle's fp32 fmaf chain on
the same bf16-representable inputs -- bit-exact; wgrad and BatchNorm within the stated tolerances."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from test_hip_core import _rb_both, dev, rand_indices

pytestmark = pytest.mark.gpu


@pytest.fixture
def fp32_weights():
    """BTC_TUNE_BF16_OPERANDS = 1: bf16 activations with FP32 weights on the fp32 MFMA (conv_apply_g) -- the bit-exact path.
    The default under bf16 activations is bf16 operands on the bf16 matrix pipe (tests/test_hip_bf16_mfma.py)."""
    from btcdet_amd._lib import check, lib
    check(lib().btc_tune_set(8, 1), "btc_tune_set")
    yield
    check(lib().btc_tune_set(8, 0), "btc_tune_set")


def _bits(t):
    return t.detach().contiguous().view(torch.int16).cpu().numpy()


def _bf16_bits(a):
    """float32 array holding bf16-representable values -> int16 bit patterns"""
    return (np.ascontiguousarray(a, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16).view(np.int16)


CASES = [(16, 16, 500), (16, 32, 500), (32, 32, 500), (32, 64, 500), (64, 64, 500), (64, 64, 9000), (32, 64, 9000), (64, 128, 400),
         (128, 128, 300), (256, 128, 300), (128, 256, 300), (48, 32, 500)]


@pytest.mark.parametrize("cin,cout,n", CASES)
@pytest.mark.parametrize("kind", ["subm", "conv"])
def test_conv_bf16_features_bit_exact(fp32_weights, cin, cout, n, kind):
    from btcdet_amd.spconv import ops
    rng = np.random.default_rng(cin * 1000 + cout + n)
    shape, B = ((12, 48, 44), 2) if n > 5000 else ((8, 20, 18), 2)
    idx = rand_indices(rng, n, B, shape)
    s = (1, 1, 1) if kind == "subm" else (2, 2, 2)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, (3, 3, 3), s, (1, 1, 1), (1, 1, 1), kind)
    feat = orc.bf16_round(rng.standard_normal((idx.shape[0], cin)).astype(np.float32))
    W = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    dout = orc.bf16_round(rng.standard_normal((o_idx.shape[0], cout)).astype(np.float32))

    f = torch.from_numpy(feat).to(dev()).to(torch.bfloat16).requires_grad_(True)
    w = torch.from_numpy(W).to(dev()).requires_grad_(True)
    bt = torch.from_numpy(bias).to(dev()).requires_grad_(True)
    out = ops.indice_conv(f, w, bt, rb)
    assert out.dtype == torch.bfloat16
    out.backward(torch.from_numpy(dout).to(dev()).to(torch.bfloat16))
    assert f.grad.dtype == torch.bfloat16 and w.grad.dtype == torch.float32

    np.testing.assert_array_equal(_bits(out), _bf16_bits(orc.bf16_round(orc.conv_fwd(feat, W, bias, o_out))))
    np.testing.assert_array_equal(_bits(f.grad), _bf16_bits(orc.bf16_round(orc.conv_dgrad(dout, W, o_in))))
    ref_dw = orc.conv_wgrad(feat, dout, o_out, W.shape)  # fp32 partial sums over row splits vs a double-precision reference
    assert np.abs(w.grad.cpu().numpy() - ref_dw).max() <= 1e-4 * (np.abs(ref_dw).max() + 1e-6)
    np.testing.assert_allclose(bt.grad.cpu().numpy(), dout.sum(0), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("cin,cout", [(6, 16), (34, 32), (32, 3)])
def test_conv_bf16_features_odd_channels_fall_back_to_fp32_compute(cin, cout, exact_conv):
    """channel counts that are not multiples of 16 compute in fp32 on the widened activations and round the result"""
    from btcdet_amd.spconv import ops
    rng = np.random.default_rng(cin + cout)
    shape, B = (8, 20, 18), 2
    idx = rand_indices(rng, 600, B, shape)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), "subm")
    feat = orc.bf16_round(rng.standard_normal((idx.shape[0], cin)).astype(np.float32))
    W = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    f = torch.from_numpy(feat).to(dev()).to(torch.bfloat16)
    out = ops.indice_conv(f, torch.from_numpy(W).to(dev()), None, rb)
    assert out.dtype == torch.bfloat16
    np.testing.assert_array_equal(_bits(out), _bf16_bits(orc.bf16_round(orc.conv_fwd(feat, W, None, o_out))))


@pytest.mark.parametrize("N,C", [(5000, 16), (777, 32), (20000, 64), (300, 128)])
def test_fused_bn_relu_bf16(N, C):
    """fused BatchNorm1d + ReLU on bf16 activations: statistics in fp64 from the bf16 inputs, output rounded to bf16.
    Reference: torch BatchNorm on the widened input; tolerance one bf16 ulp (2^-8 relative) + 1e-3 absolute"""
    from btcdet_amd.spconv import fused_bn
    torch.manual_seed(N + C)
    x = torch.randn(N, C, device=dev()).mul_(2).add_(0.5).to(torch.bfloat16)
    bn = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(dev())
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    ref_bn = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(dev())
    ref_bn.load_state_dict(bn.state_dict())
    xr = x.float().requires_grad_(True)
    yr = torch.relu(ref_bn(xr))
    g = torch.randn(N, C, device=dev()).to(torch.bfloat16)
    yr.backward(g.float())
    xb = x.clone().requires_grad_(True)
    y = fused_bn.batch_norm_relu(bn, xb, True)
    assert y.dtype == torch.bfloat16
    y.backward(g)
    torch.testing.assert_close(y.float(), yr, rtol=2 ** -7, atol=1e-3)
    # dx depends on the relu mask of the ROUNDED output only where y is within an ulp of 0: compare in aggregate
    err = (xb.grad.float() - xr.grad).abs()
    assert float(err.mean()) <= 2e-3 * float(xr.grad.abs().mean()) + 1e-5 and float(err.max()) <= 0.05 * float(xr.grad.abs().max()) + 1e-3
    torch.testing.assert_close(bn.weight.grad, ref_bn.weight.grad, rtol=2e-2, atol=2e-2 * float(ref_bn.weight.grad.abs().max()))
    torch.testing.assert_close(bn.running_mean, ref_bn.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn.running_var, ref_bn.running_var, rtol=1e-5, atol=1e-6)


def test_hot_path_bf16_features_close_to_fp32():
    """the whole hot path with FEATURE_DTYPE: bf16 in both backbones (bf16 operands on the bf16 matrix pipe where the channel
    counts allow): forward + backward run, every parameter gets a finite gradient, the occupancy branch stays within bf16
    accuracy of the fp32 run (same weights, same batch), and the detection branch -- fed the SAME merged voxels in both
    precisions, because its own input set is a top-k of the occupancy probabilities and a rounding can swap members --
    reproduces the BEV map and x_combine to a measured relative L2 error (printed; bound = 2x the observed)"""
    import bench
    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    d = dev()
    batch = bench.build_batches(1, 0, d)[0]

    def build(dtype):
        cfg = load_cfg()
        cfg.MODEL.OCC.BACKBONE_3D["FEATURE_DTYPE"] = dtype
        cfg.MODEL.BACKBONE_3D["FEATURE_DTYPE"] = dtype
        torch.manual_seed(0)
        np.random.seed(0)
        return BtcHotPath(cfg, device=d).to(d).train()

    def run(model):
        bd = model.dataset.data_processor.forward_batch(batch["points"], batch["pre_rot_points"], batch["scene_offsets"], batch["rot_z"])
        bd.update({"batch_size": 2, "points": batch["points5"], "gt_boxes": batch["gt_boxes"], "gt_boxes_num": batch["gt_boxes_num"],
                   "box_mirr_flag": batch["box_mirr_flag"], "bm_points": batch["bm_points"], "rot_z": batch["rot_z"], "is_train": True})
        ret, tb, out = model(bd)
        loss = ret["loss_occ"] + 1e-3 * ret["spatial_features"].pow(2).mean() + 1e-3 * ret["x_combine"].float().pow(2).mean()
        loss.backward()
        torch.cuda.synchronize()
        grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad}
        det_in = {k: out[k].detach().clone() for k in ("voxels", "voxel_num_points", "voxel_coords")}
        return float(ret["loss_occ"].detach()), grads, out["batch_pred_occ_prob"].detach(), det_in

    def det(model, det_in):
        with torch.no_grad():
            b = dict(det_in)
            b["batch_size"] = 2
            for mod in model.det_module_list:
                b = mod(b)
        return b["spatial_features"].float(), b["multi_scale_3d_features"]["x_combine"].features.float()

    m32, m16 = build("fp32"), build("bf16")
    l32, g32, p32, det_in = run(m32)
    l16, g16, p16, _ = run(m16)
    assert all(g is not None and torch.isfinite(g).all() for g in g16.values())
    print("occupancy loss fp32 %.6f bf16 %.6f; |dp| max %.2e mean %.2e" % (l32, l16, float((p16 - p32).abs().max()), float((p16 - p32).abs().mean())))
    assert abs(l16 - l32) <= 2e-2 * abs(l32) + 1e-3
    assert float((p16 - p32).abs().max()) < 3e-2 and float((p16 - p32).abs().mean()) < 2e-3
    bev32, xc32 = det(m32, det_in)
    bev16, xc16 = det(m16, det_in)
    rel_bev = float((bev16 - bev32).norm() / bev32.norm())
    rel_xc = float((xc16 - xc32).norm() / xc32.norm())
    print("detection branch on identical inputs, bf16 vs fp32: BEV rel L2 %.3e, x_combine rel L2 %.3e" % (rel_bev, rel_xc))
    assert rel_bev < 3e-2 and rel_xc < 3e-2, (rel_bev, rel_xc)     # observed ~1.2e-2 (20 layers of bf16 rounding + train-mode BatchNorm)
