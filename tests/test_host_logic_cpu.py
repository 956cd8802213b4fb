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


def test_group_optimizer_reads_gradients_from_fixed_buffers():
    """btcdet_amd.train_step.GroupOptimizer (clip + decoupled decay + fused Adam + OneCycle; the arithmetic is pinned to the
    reference by tests/test_train_step_cpu.py): taking the gradients from a reducer's fixed buffers == taking them from param.grad"""
    from btcdet_amd.train_step import GroupOptimizer

    def mk():
        torch.manual_seed(0)
        return [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(11))], [torch.nn.Parameter(torch.randn(3, 3, 4))]
    (a1, a2), (b1, b2) = mk(), mk()
    groups = lambda g1, g2: [{"params": g1, "lr": 3e-3, "weight_decay": 0.001, "grad_norm_clip": 10.0},
                             {"params": g2, "lr": 1e-2, "weight_decay": 0.01, "grad_norm_clip": 10.0}]
    try:
        ref, lean = GroupOptimizer(groups(a1, a2), 100), GroupOptimizer(groups(b1, b2), 100)
        bufs = {id(p): torch.zeros_like(p) for p in b1 + b2}
        for it in range(5):
            torch.manual_seed(10 + it)
            gs = [torch.randn_like(p) * (30.0 if it == 2 else 1.0) for p in a1 + a2]
            for p, q, g in zip(a1 + a2, b1 + b2, gs):
                p.grad = g.clone()
                if it < 3:
                    q.grad = g.clone()
                else:                      # gradients come from a reducer's flat buffers
                    q.grad = None
                    bufs[id(q)].copy_(g)
            if it == 3:
                lean.read_grads_from(lambda p: bufs[id(p)])
            ref.step()
            lean.step()
    except (RuntimeError, NotImplementedError) as e:  # a torch build without the fused CPU kernel
        pytest.skip("torch._fused_adam_ unavailable on CPU here: %s" % e)
    for p, q in zip(a1 + a2, b1 + b2):
        torch.testing.assert_close(q, p, rtol=0, atol=0)


def test_geometry_plans_of_both_backbones():
    """spconv/geometry.py: which layer builds, which reuses (by indice_key, or through an identical geometry on the same level) and
    the shapes along the chain -- derived on the host, no GPU needed"""
    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    from btcdet_amd.spconv.geometry import GeometryPlan, flatten_convs
    m = BtcHotPath(load_cfg(), device="cpu")
    bb, head = m.occ_modules.backbone_3d, m.occ_modules.occ_dense_head
    convs = flatten_convs(bb.conv1, bb.conv2, bb.conv3, bb.deconv4, bb.deconv5, head.conv_cls, head.conv_res)
    plan = GeometryPlan(convs, bb.sparse_shape, 2)
    kinds = [e[0] for e in plan.entries]
    assert kinds == [1, 1, 0, 1, 0, 1, 0, 1, 0, 3, 3]                     # 9 builds, the two head convs reuse subm5's rulebook
    assert [e[1] for e in plan.entries][-2:] == [8, 8]
    assert plan.entries[1][4] == [5, 79, 105] and plan.entries[3][4] == [3, 40, 53] and plan.entries[7][4] == [9, 157, 209]
    assert len(plan.args) == 11 and all(len(a) == len(convs) for a in plan.args)
    det = m.det_modules.backbone_3d
    stages = [det.conv1, det.conv2, det.conv2_combine, det.conv3, det.conv3_combine, det.conv4, det.conv4_combine, det.conv_out]
    if getattr(det, "squeezeBev", None) is not None:
        stages.append(det.squeezeBev)
    dplan = GeometryPlan(flatten_convs(*stages), det.sparse_shape, 2)
    dk = [(c.indice_key, e[0]) for c, e in zip(dplan.convs, dplan.entries)]
    assert dk[:10] == [("subm1", 0), ("spconv2", 1), ("subm2", 0), ("subm2", 3), ("spconv3", 1), ("subm3", 0), ("subm3", 3), ("spconv4", 1),
                       ("subm4", 0), ("subm4", 3)]
    assert dplan.entries[10][4] == [2, 200, 176]                          # conv_out: (3,1,1) / stride (2,1,1) on [5, 200, 176]
    with pytest.raises(TypeError):
        flatten_convs(torch.nn.ReLU())
