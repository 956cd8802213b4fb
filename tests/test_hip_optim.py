"""GroupOptimizer's flat path (csrc/optim.hip: norm clip + decoupled decay + Adam of a whole parameter group in three launches over
flat buffers) against its list path (torch multi-tensor ops; pinned to the reference's OptimWrapper / clip_grad_norm_ / OneCycle by
tests/test_train_step_cpu.py): same parameters and moments to fp32 rounding, same schedule, same checkpoint state; parameter
version counters move (the bf16 weight copies are keyed on them); a step with a missing gradient falls back to the list path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _nets(seed):
    torch.manual_seed(seed)
    # (no BatchNorm behind a biased Linear here: that bias has a zero gradient up to rounding noise, which Adam's m / sqrt(v)
    # normalises to O(1) -- two correct implementations then differ by whole update steps)
    a = torch.nn.Sequential(torch.nn.Linear(37, 129), torch.nn.Tanh(), torch.nn.Linear(129, 5)).cuda()
    b = torch.nn.Sequential(torch.nn.Linear(5, 2048), torch.nn.Linear(2048, 3)).cuda()   # a parameter of more than one 1024-element chunk
    return a, b


def _opt(a, b, flat):
    from btcdet_amd.train_step import GroupOptimizer
    kw = dict(moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4, lr_clip=1e-7)
    return GroupOptimizer([dict(params=list(a.parameters()), lr=0.003, weight_decay=0.001, grad_norm_clip=0.05, **kw),   # clip bites
                           dict(params=list(b.parameters()), lr=0.01, weight_decay=0.01, grad_norm_clip=1e9, **kw)], total_steps=30, flat=flat)


def _run(flat, steps=8, drop_grad_at=None):
    a, b = _nets(0)
    opt = _opt(a, b, flat)
    assert ("flat" in opt.groups[0]) == flat
    torch.manual_seed(1)
    versions = [p._version for p in a.parameters()]
    for it in range(steps):
        x = torch.randn(16, 37, device="cuda")
        opt.zero_grad()
        b(a(x)).square().mean().backward()
        if drop_grad_at == it:
            b[1].bias.grad = None
        opt.step()
    assert all(p._version > v for p, v in zip(a.parameters(), versions))
    return a, b, opt


def test_flat_path_equals_list_path():
    a0, b0, o0 = _run(False)
    a1, b1, o1 = _run(True)
    assert o0.iteration == o1.iteration == 8 and o0.lrs() == o1.lrs()
    for p0, p1 in zip(list(a0.parameters()) + list(b0.parameters()), list(a1.parameters()) + list(b1.parameters())):
        np.testing.assert_allclose(p1.detach().cpu().numpy(), p0.detach().cpu().numpy(), rtol=2e-5, atol=2e-7)
    for s0, s1 in zip(o0.state_dict_lst(), o1.state_dict_lst()):
        assert s0["param_groups"] == s1["param_groups"] and set(s0["state"]) == set(s1["state"])
        for k in s0["state"]:
            assert float(s0["state"][k]["step"]) == float(s1["state"][k]["step"]) == 8.0
            for name in ("exp_avg", "exp_avg_sq"):
                ref = s0["state"][k][name].cpu().numpy()   # elements near zero are sums that cancel: tolerance relative to the tensor's scale
                np.testing.assert_allclose(s1["state"][k][name].cpu().numpy(), ref, rtol=2e-5, atol=2e-6 * float(np.abs(ref).max()))


def test_parameters_are_views_of_the_flat_buffer_and_checkpoints_round_trip():
    a, b, opt = _run(True, steps=3)
    fl = opt.groups[0]["flat"]
    off = 0
    for p in a.parameters():
        assert p.data_ptr() == fl["p"].data_ptr() + 4 * off
        off += p.numel()
    states = opt.state_dict_lst()
    a2, b2 = _nets(0)
    opt2 = _opt(a2, b2, True)
    for q, p in zip(list(a2.parameters()) + list(b2.parameters()), list(a.parameters()) + list(b.parameters())):
        q.data.copy_(p.data)
    opt2.load_state_dict_lst(states, iteration=opt.iteration)
    torch.manual_seed(7)
    x = torch.randn(16, 37, device="cuda")
    for o, (m, n) in ((opt, (a, b)), (opt2, (a2, b2))):
        o.zero_grad()
        n(m(x)).square().mean().backward()
        o.step()
    for q, p in zip(list(a2.parameters()) + list(b2.parameters()), list(a.parameters()) + list(b.parameters())):
        assert torch.equal(q, p)


def test_missing_gradient_takes_the_list_path():
    a0, b0, o0 = _run(False, drop_grad_at=3)
    a1, b1, o1 = _run(True, drop_grad_at=3)
    assert o1.groups[1]["flat"]["n"] == -1 and o1.groups[0]["flat"]["n"] == 8
    for p0, p1 in zip(list(a0.parameters()) + list(b0.parameters()), list(a1.parameters()) + list(b1.parameters())):
        np.testing.assert_allclose(p1.detach().cpu().numpy(), p0.detach().cpu().numpy(), rtol=2e-5, atol=2e-7)
    s0, s1 = o0.state_dict_lst()[1], o1.state_dict_lst()[1]
    assert [float(s0["state"][k]["step"]) for k in sorted(s0["state"])] == [float(s1["state"][k]["step"]) for k in sorted(s1["state"])]


@pytest.mark.parametrize("tag,lr,wd", [("det", 0.01, 0.01), ("occ", 0.003, 0.001)])
def test_flat_path_on_the_gpu_vs_the_reference_loop(tag, lr, wd):
    """the three-launch flat optimizer step ON THE GPU against the vectors the reference's own OptimWrapper / OneCycle /
    clip_grad_norm_ loop body wrote (tests/golden/gen_optim_golden.py -> optim.npz; tests/test_train_step_cpu.py checks the list path
    on the CPU against the same file): lr / beta1 sequence exactly, parameters after 1, 2, 3, 16-18, 40 and 43 steps, norm clip biting on
    every third step"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import common
    from test_train_step_cpu import tiny
    from btcdet_amd.train_step import GroupOptimizer
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optim.npz"))
    per_epoch, epochs = [int(v) for v in g["meta"]]
    m = tiny().cuda()
    opt = GroupOptimizer([dict(params=list(m.parameters()), lr=lr, weight_decay=wd, grad_norm_clip=10.0, moms=(0.95, 0.85), div_factor=10,
                               pct_start=0.4, lr_clip=1e-7)], total_steps=per_epoch * epochs, flat=True)
    assert "flat" in opt.groups[0]
    snaps = {int(i): k for k, i in enumerate(g["snap_iters"])}
    for it in range(per_epoch * epochs + 3):
        assert opt.groups[0]["lr"] == pytest.approx(float(g[tag + "_lr"][it]), rel=1e-12, abs=0)
        assert opt.groups[0]["mom"] == pytest.approx(float(g[tag + "_mom"][it]), rel=1e-12, abs=0)
        opt.zero_grad()
        for j, (n, p) in enumerate(m.named_parameters()):
            u = torch.from_numpy(common._hash01(p.numel(), 100 * it + j)).reshape(p.shape)
            p.grad = ((u - 0.5) * (40.0 if it % 3 == 0 else 0.5)).cuda()
        opt.step()
        if it in snaps:
            got = np.concatenate([p.detach().cpu().numpy().reshape(-1) for p in m.parameters()])
            np.testing.assert_allclose(got, g[tag + "_params"][snaps[it]], rtol=2e-5, atol=2e-7, err_msg="after step %d" % it)
