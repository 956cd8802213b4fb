"""host-side pieces that need no GPU: LazyScalar (the loss scalars OccHead3D.get_loss hands out instead of .item()), the
layer-chain planner's fallbacks, the stand-in modules' CPU refusal (no CPU fallback in the product path)."""
import numpy as np
import pytest
import torch


class _Ev(object):
    def __init__(self):
        self.waited = 0

    def synchronize(self):
        self.waited += 1


def test_lazy_scalar_reads_wait_once_and_behave_like_floats():
    from btcdet_amd.occ_head import LazyScalar, _lazy_scalars
    ev = _Ev()
    host = torch.tensor([1.5, -2.25])
    a, b = LazyScalar(host, 0, ev), LazyScalar(host, 1, ev)
    assert ev.waited == 0                                  # nothing waits until a value is read
    assert float(a) == 1.5 and ev.waited == 1
    assert a.item() == 1.5 and ev.waited == 1              # cached
    assert "%.2f" % a == "1.50" and "{:.1f}".format(b) == "-2.2" and repr(a) == "1.5"
    assert a + 1 == 2.5 and 1 + a == 2.5 and a - b == 3.75 and 2 * a == 3.0 and a / 3 == 0.5 and -b == 2.25 and abs(b) == 2.25
    assert a > b and b < 0 and a >= 1.5 and a == 1.5 and bool(a)
    np.testing.assert_allclose([a, b], [1.5, -2.25])       # numpy converts through __float__
    assert _lazy_scalars(torch.tensor([3.0, 4.0, 5.0]), 2) == [3.0, 4.0]   # CPU tensors: plain floats


def test_chain_planner_falls_back_off_gpu_and_on_foreign_modules():
    import btcdet_amd.spconv as spconv
    from functools import partial
    from btcdet_amd.backbones_3d import post_act_block
    norm = partial(torch.nn.BatchNorm1d, eps=1e-3, momentum=0.01)
    seq = spconv.SparseSequential(post_act_block(4, 8, 3, norm_fn=norm, indice_key="a"), post_act_block(8, 8, 3, norm_fn=norm, indice_key="a"))
    assert [type(m).__name__ for m in seq._flat_modules()] == ["SubMConv3d", "BatchNorm1d", "ReLU"] * 2
    idx = torch.tensor([[0, 1, 1, 1], [0, 1, 1, 2]], dtype=torch.int32)
    x = spconv.SparseConvTensor(torch.randn(2, 4), idx, [4, 4, 4], 1)
    assert seq._chain_plan(x) is None                      # CPU tensor: the layer-by-layer path decides what to do
    assert spconv.SparseSequential(torch.nn.ReLU())._chain_plan(x) is None


def test_op_stand_ins_refuse_cpu_tensors():
    from btcdet_amd import pointnet2_stack as p2
    xyz, cnt = torch.zeros(4, 3), torch.tensor([4], dtype=torch.int32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        p2.ball_query(1.0, 4, xyz, cnt, xyz, cnt)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        p2.furthest_point_sample(torch.zeros(1, 8, 3), 2)
