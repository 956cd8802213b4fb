"""Backward statistics of a BatchNorm (+ ReLU) gathered in the epilogue of the data gradient that produces its dL/dy (csrc/bn_fuse.h `bwd`,
btc_conv_dgrad_bn_bwd; round 6) -- the reference's post_act_block chains (spconv_backbone.py:33-43), where a block's output feeds the next
block and nothing else.

  * C ABI: din has the bits of the plain data gradient (btc_conv_apply_src); dgamma / dbeta equal the statistics of btc_bn_relu_bwd's first
    launch on the same din to 1e-6 of their scale (fp64 sums in another order) and a float64 torch product to 2e-6; the z-split path (few
    rows: partial slabs + split_reduce) and the PAIR items (32-channel reductions) included; BTC_TUNE_BN_BWD_FUSE = 1 says `fused = 0` and
    leaves dgamma / dbeta untouched; btc_bn_relu_bwd_apply with those statistics == btc_bn_relu_bwd's dx bit for bit.
  * module level: a SparseSequential of conv -> BatchNorm1d -> ReLU triples run as ONE chain (conv_bn_relu_chain): every gradient with the
    links equals the run without them (set_bn_bwd_fuse(False)) to 2e-6 of its scale, twice in a row (the ring of links is reused), and
    a chain whose middle output is ALSO read by somebody else is not a chain (the per-module path: no links, same numbers).
"""
import ctypes

import numpy as np
import pytest
import torch

from test_hip_core import _rb_both, dev, rand_indices

pytestmark = pytest.mark.gpu
FUSE_KEY = 23   # BTC_TUNE_BN_BWD_FUSE


def _setup(rng, cin, cout, n_vox, kind="subm"):
    from btcdet_amd import _lib
    from btcdet_amd._lib import check, ptr, stream_ptr
    L = _lib.lib()
    shape, B = (12, 48, 44), 2
    idx = rand_indices(rng, n_vox, B, shape)
    s = (1, 1, 1) if kind == "subm" else (2, 2, 2)
    (o_idx, o_out, o_in, o_sh), rb = _rb_both(idx, B, shape, (3, 3, 3), s, (1, 1, 1), (1, 1, 1), kind)
    n_in, n_out = idx.shape[0], o_idx.shape[0]
    w = torch.from_numpy((rng.standard_normal((27, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)).to(dev())
    q = torch.empty((2, 3 * w.numel()), dtype=torch.bfloat16, device=dev())
    check(L.btc_weights_split3(ptr(w), 27, cin, cout, ptr(q[0]), ptr(q[1]), stream_ptr()), "btc_weights_split3")
    dout = torch.from_numpy(rng.standard_normal((n_out, cout)).astype(np.float32)).to(dev())
    # the BatchNorm in front: x its input (this conv's input level), y = relu(gamma xhat + beta)
    x = torch.from_numpy((rng.standard_normal((n_in, cin)) * 1.7 + 0.3).astype(np.float32)).to(dev())
    mean, var = x.double().mean(0), x.double().var(0, unbiased=False)
    stats = torch.stack([mean, 1.0 / torch.sqrt(var + 1e-3)]).float().contiguous()
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, cin).astype(np.float32)).to(dev())
    beta = torch.from_numpy(rng.uniform(-0.5, 0.5, cin).astype(np.float32)).to(dev())
    y = torch.relu((x - stats[0]) * stats[1] * gamma + beta).contiguous()
    return L, rb, q, dout, x, y, stats, gamma, n_in, n_out


@pytest.mark.parametrize("cin,cout,n_vox,kind", [(64, 64, 9000, "subm"), (32, 32, 30000, "subm"), (32, 64, 30000, "subm"), (64, 32, 12000, "subm"),
                                                 (128, 64, 3000, "subm"), (64, 64, 3000, "subm"), (128, 128, 9000, "subm"), (32, 64, 40000, "conv")])
def test_dgrad_with_bn_backward_statistics(cin, cout, n_vox, kind):
    from btcdet_amd._lib import check, ptr, stream_ptr
    rng = np.random.default_rng(cin * 5 + cout + n_vox)
    L, rb, q, dout, x, y, stats, gamma, n_in, n_out = _setup(rng, cin, cout, n_vox, kind)
    scratch = torch.zeros((48 << 20,), dtype=torch.uint8, device=dev())
    check(L.btc_set_scratch(stream_ptr(), ptr(scratch), scratch.numel()), "btc_set_scratch")     # small levels run z-split
    try:
        mirror = rb.mirrored
        pass_, m = (2, rb.nbr_out) if mirror else (1, rb.map_bwd)
        order = None if mirror else rb.order_in
        plain = torch.empty((n_in, cin), device=dev())
        check(L.btc_conv_apply_src(pass_, 3, ptr(dout), n_out, ptr(q[0]), None, ptr(m), ptr(order), n_in, 27, cin, cout, ptr(plain), stream_ptr()), "dgrad")
        fw = torch.zeros((L.btc_bn_fuse_ws_bytes(),), dtype=torch.uint8, device=dev())
        for rep in range(2):        # (the slots and the counter are left zeroed: a second launch on the same workspace)
            din = torch.full((n_in, cin), float("nan"), device=dev())
            dparam = torch.full((2, cin), float("nan"), device=dev())
            fused = ctypes.c_int(-1)
            check(L.btc_conv_dgrad_bn_bwd(pass_, ptr(dout), n_out, ptr(q[0]), ptr(m), ptr(order), n_in, 27, cin, cout, ptr(din), ptr(x), ptr(y), ptr(stats[0]),
                                          ptr(stats[1]), 1, ptr(dparam[0]), ptr(dparam[1]), ptr(fw), stream_ptr(), ctypes.byref(fused)), "fused dgrad")
            assert fused.value == 1
            assert torch.equal(din, plain)
            g = torch.where(y > 0, din, torch.zeros_like(din)).double()
            xh = ((x - stats[0]) * stats[1]).double()
            ref_b, ref_g = g.sum(0), (g * xh).sum(0)
            sb, sg = float(ref_b.abs().max()) + 1e-30, float(ref_g.abs().max()) + 1e-30
            assert float((dparam[1].double() - ref_b).abs().max()) <= 2e-6 * sb and float((dparam[0].double() - ref_g).abs().max()) <= 2e-6 * sg
            assert int(fw.view(torch.int32)[0]) == 0 and float(fw[256:].view(torch.float64).abs().max()) == 0.0
        # the two launches of btc_bn_relu_bwd on the same din: the statistics to 1e-6, dx from the fused statistics == its dx where they agree
        wsb = L.btc_bn_ws_bytes(cin)
        ws = torch.zeros((wsb,), dtype=torch.uint8, device=dev())
        dx0, dp0 = torch.empty_like(x), torch.empty((2, cin), device=dev())
        check(L.btc_bn_relu_bwd(ptr(x), ptr(y), ptr(din), n_in, cin, ptr(gamma), ptr(stats[0]), ptr(stats[1]), 1, 1, ptr(dx0), ptr(dp0[0]), ptr(dp0[1]), ptr(ws), wsb,
                                stream_ptr()), "bn bwd")
        assert float((dp0 - dparam).abs().max()) <= 1e-6 * float(dp0.abs().max())
        dx1 = torch.empty_like(x)
        check(L.btc_bn_relu_bwd_apply(0, ptr(x), ptr(y), ptr(din), n_in, cin, ptr(gamma), ptr(stats[0]), ptr(stats[1]), 1, 1, ptr(dx1), ptr(dp0[0]), ptr(dp0[1]),
                                      stream_ptr()), "bn bwd apply")
        assert torch.equal(dx1, dx0)
        # switched off: din only, the statistics untouched
        assert L.btc_tune_set(FUSE_KEY, 1) == 0
        try:
            dparam.fill_(7.0)
            fused = ctypes.c_int(-1)
            check(L.btc_conv_dgrad_bn_bwd(pass_, ptr(dout), n_out, ptr(q[0]), ptr(m), ptr(order), n_in, 27, cin, cout, ptr(din), ptr(x), ptr(y), ptr(stats[0]),
                                          ptr(stats[1]), 1, ptr(dparam[0]), ptr(dparam[1]), ptr(fw), stream_ptr(), ctypes.byref(fused)), "fused dgrad off")
            assert fused.value == 0 and torch.equal(din, plain) and bool((dparam == 7.0).all())
        finally:
            assert L.btc_tune_set(FUSE_KEY, 0) == 0
    finally:
        check(L.btc_set_scratch(stream_ptr(), None, 0), "btc_set_scratch")
        torch.cuda.synchronize()


def _chain(rng, chans, n_vox):
    from btcdet_amd import spconv
    from torch import nn
    mods = []
    for i, (a, b) in enumerate(zip(chans[:-1], chans[1:])):
        mods += [spconv.SubMConv3d(a, b, 3, padding=1, bias=False, indice_key="subm_t"), nn.BatchNorm1d(b, eps=1e-3, momentum=0.01), nn.ReLU()]
    net = spconv.SparseSequential(*mods).to(dev()).train()
    shape, B = (12, 48, 44), 2
    idx = torch.from_numpy(rand_indices(rng, n_vox, B, shape)).to(dev())
    feat = torch.from_numpy(rng.standard_normal((idx.shape[0], chans[0])).astype(np.float32)).to(dev())
    return net, idx, feat, shape, B


def _grads(net, idx, feat, shape, B, tap=None):
    from btcdet_amd import spconv
    f = feat.clone().requires_grad_(True)
    x = spconv.SparseConvTensor(f, idx, list(shape), B)
    net.forward_geometry(x)          # the rulebooks in place: the container runs as ONE chain (SparseSequential._chain_plan)
    for p in net.parameters():
        p.grad = None
    y = net(x)
    loss = (y.features * y.features).sum() * 1e-3 + y.features.sum()
    loss.backward()
    from btcdet_amd.spconv import ops
    ops.join_wgrad()
    torch.cuda.synchronize()
    return [f.grad.clone()] + [p.grad.clone() for p in net.parameters()]


@pytest.mark.parametrize("chans,n_vox", [((32, 32, 64, 64), 30000), ((64, 64, 64), 9000), ((64, 128, 64, 32), 3000)])
def test_chain_gradients_with_and_without_links(chans, n_vox):
    from btcdet_amd import _lib
    F = _lib.fast()
    if F is None:
        pytest.skip("the chain is driven by the compiled binding")
    rng = np.random.default_rng(sum(chans) + n_vox)
    net, idx, feat, shape, B = _chain(rng, chans, n_vox)
    # the chain needs its rulebooks in place: one warm call builds them (indice_key)
    F.set_bn_bwd_fuse(False)
    try:
        hits = F.bn_bwd_fuse_hits()
        ref = _grads(net, idx, feat, shape, B)
        ref2 = _grads(net, idx, feat, shape, B)
        assert F.bn_bwd_fuse_hits() == hits
    finally:
        F.set_bn_bwd_fuse(True)
    for a, b in zip(ref, ref2):
        assert torch.equal(a, b)
    for rep in range(2):
        hits = F.bn_bwd_fuse_hits()
        got = _grads(net, idx, feat, shape, B)
        assert F.bn_bwd_fuse_hits() - hits == len(chans) - 2, "every layer but the last hands its BatchNorm's statistics to the next one's data gradient"
        for a, b in zip(got, ref):
            scale = float(b.abs().max()) + 1e-30
            assert bool(torch.isfinite(a).all()) and float((a - b).abs().max()) <= 2e-6 * scale
