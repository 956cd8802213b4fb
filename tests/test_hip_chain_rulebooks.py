"""Every rulebook of a CHAIN of sparse layers (btc_chain_levels / btc_chain_maps, csrc/rulebook.hip) against the oracle, map by map:
round 3 builds a strided layer's nbr_out by a gather from its output rows (no -1 fill, no scatter), keeps ONE map per submanifold
layer (nbr_in is its mirror image) and probes through LDS-staged bitmap windows where a workgroup's rows lie in one plane -- on a
small dense grid most workgroups stage, on the KITTI detection grid none does, the KITTI occupancy grid mixes both; the results must
not depend on which path a workgroup took.  Also: dgrad through the mirrored forward map == dgrad through the explicit backward map."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402  (the checker)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (kernel, stride, padding, mode, key): the occupancy backbone's layer pattern (spconv_backbone.py:106-128) + a (3,1,1) strided tail
CHAIN = [(3, 1, 1, "conv", "c1"), (3, 2, 1, "conv", "c2"), (3, 1, 0, "subm", "s2"), (3, 2, 1, "conv", "c3"), (3, 1, 0, "subm", "s3"),
         (3, 2, 1, "transpose", "d4"), (3, 1, 0, "subm", "s4"), (3, 2, 1, "transpose", "d5"), (3, 1, 0, "subm", "s5"),
         ((3, 1, 1), (2, 1, 1), 0, "conv", "tail")]


def _net(chain):
    from btcdet_amd import spconv
    layers = []
    for k, s, p, mode, key in chain:
        if mode == "subm":
            layers.append(spconv.SubMConv3d(4, 4, k, bias=False, indice_key=key))
        elif mode == "conv":
            layers.append(spconv.SparseConv3d(4, 4, k, stride=s, padding=p, bias=False, indice_key=key))
        else:
            layers.append(spconv.SparseConvTranspose3d(4, 4, k, stride=s, padding=p, bias=False, indice_key=key))
    return spconv.SparseSequential(*layers)


def _random_indices(rng, n, B, shape):
    cells = rng.choice(B * shape[0] * shape[1] * shape[2], size=n, replace=False)
    rng.shuffle(cells)   # arbitrary (voxelizer-like) order: the chain's input level is hashed, not ranked
    vol = shape[0] * shape[1] * shape[2]
    b, r = cells // vol, cells % vol
    return np.stack([b, r // (shape[1] * shape[2]), (r // shape[2]) % shape[1], r % shape[2]], 1).astype(np.int32)


def _check_chain(idx, B, shape, chain):
    from btcdet_amd.spconv.geometry import GeometryPlan, flatten_convs
    net = _net(chain)
    plan = GeometryPlan(flatten_convs(net), shape, B)
    indice_dict = {}
    rbs = plan.run(torch.from_numpy(idx).to(DEV), indice_dict)
    cur, sh = idx, list(shape)
    modes = {"conv": orc.MODE_CONV, "transpose": orc.MODE_TRANSPOSE, "subm": orc.MODE_SUBM}
    for (k, s, p, mode, key), rb in zip(chain, rbs):
        o_idx, o_out, o_in, o_sh = orc.rulebook(cur, sh, k, s, p, 1, modes[mode])
        assert list(rb.out_shape) == list(o_sh), key
        np.testing.assert_array_equal(rb.out_indices.cpu().numpy(), o_idx, err_msg=key)
        np.testing.assert_array_equal(rb.nbr_out.cpu().numpy(), o_out, err_msg=key + " nbr_out")
        np.testing.assert_array_equal(rb.nbr_in.cpu().numpy(), o_in, err_msg=key + " nbr_in")
        assert rb.mirrored == (mode == "subm")      # one map per submanifold layer, two per strided layer
        cur, sh = o_idx, list(o_sh)
    return rbs


@pytest.fixture(params=[0, 1], ids=["composed_marks", "one_launch_per_level"])
def mark_mode(request):
    """BTC_TUNE_RB_MARK_MULTI: the leading run of strided conv layers marked by ONE launch from the input rows (boxes composed per axis,
    round 5), or every level by its own launch -- both must give the oracle's levels"""
    from btcdet_amd._lib import lib
    assert lib().btc_tune_set(19, request.param) == 0
    yield request.param
    lib().btc_tune_set(19, 0)


@pytest.mark.parametrize("shape,n,B", [((9, 40, 60), 9000, 2), ((9, 40, 60), 300, 3), ((5, 12, 300), 4000, 2), ((17, 64, 64), 20000, 1)])
def test_chain_rulebooks_small_grids(shape, n, B, mark_mode):
    rng = np.random.default_rng(n + shape[2])
    _check_chain(_random_indices(rng, n, B, shape), B, shape, CHAIN)


# encoder runs the composed marks cover end to end: stride 1 / 2 / 3, even kernels, asymmetric padding, a (3, 1, 1) tail, cells on every
# face of the grid (a full small grid), a run that outgrows the box bound (four stride-1 layers: the last is marked from its bitmap)
ENCODERS = [
    [(3, 2, 1, "conv", "a"), (3, 2, 1, "conv", "b"), (3, 2, (0, 1, 1), "conv", "c"), ((3, 1, 1), (2, 1, 1), 0, "conv", "d")],
    [(2, 2, 0, "conv", "a"), (3, 1, 1, "conv", "b"), ((3, 3, 2), (3, 2, 2), (1, 0, 1), "conv", "c")],
    [(3, 1, 1, "conv", "a"), (3, 1, 0, "conv", "b"), (3, 1, 1, "conv", "c"), (3, 1, 1, "conv", "d")],
    [(3, 2, 1, "conv", "a"), (3, 1, 0, "subm", "s"), (3, 2, 1, "conv", "b"), (3, 2, 1, "transpose", "u"), (3, 2, 1, "conv", "c")],
]


@pytest.mark.parametrize("chain", ENCODERS)
@pytest.mark.parametrize("shape,n,B", [((25, 31, 45), 5000, 2), ((27, 6, 5), 120, 1), ((29, 20, 70), 40, 3)])
def test_encoder_runs(chain, shape, n, B, mark_mode):
    rng = np.random.default_rng(n + len(chain))
    _check_chain(_random_indices(rng, n, B, shape), B, shape, chain)


def test_chain_rulebooks_kitti_occupancy_and_detection_grids(mark_mode):
    from btcdet_amd import synth
    b = synth.make_batch([31, 32])
    og = orc.VoxelGeneratorV2(synth.KITTI_OCC_VOXEL, synth.KITTI_OCC_RANGE, 12, 20000)
    dg = orc.VoxelGeneratorV2(synth.KITTI_DET_VOXEL, synth.KITTI_DET_RANGE, 5, 16000)
    occ = np.concatenate([np.pad(og.generate(orc.absxyz_2_cylinxyz_np(s["pre_rot_points"]))["coordinates"], ((0, 0), (1, 0)), constant_values=i)
                          for i, s in enumerate(b["scenes"])]).astype(np.int32)
    det = np.concatenate([np.pad(dg.generate(s["points"])["coordinates"], ((0, 0), (1, 0)), constant_values=i)
                          for i, s in enumerate(b["scenes"])]).astype(np.int32)
    _check_chain(occ, 2, (9, 157, 209), CHAIN[:9])
    det_chain = [(3, 1, 0, "subm", "s1"), (3, 2, 1, "conv", "c2"), (3, 1, 0, "subm", "s2"), (3, 2, 1, "conv", "c3"), (3, 1, 0, "subm", "s3"),
                 (3, 2, (0, 1, 1), "conv", "c4"), (3, 1, 0, "subm", "s4"), ((3, 1, 1), (2, 1, 1), 0, "conv", "out")]
    _check_chain(det, 2, (41, 1600, 1408), det_chain)


@pytest.mark.parametrize("cin,cout,dtype", [(4, 16, "f32"), (16, 16, "f32"), (32, 3, "f32"), (64, 64, "f32"), (20, 150, "f32"), (64, 64, "bf16"), (32, 64, "bf16")])
def test_dgrad_through_the_mirrored_map_equals_dgrad_through_the_explicit_map(cin, cout, dtype):
    """BTC_PASS_DGRAD_MIRROR on nbr_out == BTC_PASS_DGRAD on the materialised nbr_in, bit for bit, in every kernel family the
    dispatch can pick (register-staged, weight-stationary, LDS-DMA fp32, bf16 operands)"""
    from btcdet_amd import _lib
    from btcdet_amd.spconv import ops
    L, ptr = _lib.lib(), _lib.ptr
    rng = np.random.default_rng(cin * 7 + cout)
    shape, B = (8, 20, 18), 2
    idx = _random_indices(rng, 2500, B, shape)
    rb = ops.build_rulebook(torch.from_numpy(idx).to(DEV), B, shape, 3, 1, 1, 1, 0, True, False)
    assert rb.mirrored and rb.map_bwd is rb.nbr_out
    n, K = rb.nbr_out.shape
    w = torch.from_numpy((rng.standard_normal((K, cin, cout)) / np.sqrt(cin)).astype(np.float32)).to(DEV)
    dout = torch.from_numpy(rng.standard_normal((n, cout)).astype(np.float32)).to(DEV)
    operands = 0
    wq = w
    if dtype == "bf16":
        dout = dout.to(torch.bfloat16)
        if L.btc_conv_bf16w_supported(K, cout, cin) == 1:
            operands = 2
            q = torch.empty((2, w.numel()), dtype=torch.bfloat16, device=DEV)
            _lib.check(L.btc_weights_to_bf16(ptr(w), K, cin, cout, ptr(q[0]), ptr(q[1]), _lib.stream_ptr()), "btc_weights_to_bf16")
            wq = q[0]
        else:
            operands = 1
    outs = []
    for pass_, nbr in ((1, rb.nbr_in), (2, rb.nbr_out)):
        din = torch.full((n, cin), float("nan"), dtype=dout.dtype, device=DEV)
        _lib.check(L.btc_conv_apply_ordered(pass_, operands, ptr(dout), ptr(wq), None, ptr(nbr), None, n, K, cin, cout, ptr(din), _lib.stream_ptr()),
                   "btc_conv_apply_ordered")
        outs.append(din)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    if dtype == "f32":
        np.testing.assert_array_equal(outs[1].cpu().numpy(), orc.conv_dgrad(dout.cpu().numpy(), w.cpu().numpy().reshape(3, 3, 3, cin, cout), rb.nbr_in.cpu().numpy()))
