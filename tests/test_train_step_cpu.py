"""btcdet_amd.train_step (norm clip + decoupled weight decay + fused Adam + OneCycle) against the reference's OWN
OptimWrapper / OneCycle / clip_grad_norm_ loop body (tests/golden/gen_optim_golden.py -> optim.npz): the lr / beta1 sequence
exactly, the parameters after 1, 2, 3, 16-18, 40 and 43 steps to float32 rounding."""
import os
import sys

import numpy as np
import pytest
import torch
from torch import nn

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import common  # noqa: E402

from btcdet_amd.train_step import GroupOptimizer, OneCycle  # noqa: E402


def tiny():
    m = nn.Sequential(nn.Linear(6, 8), nn.BatchNorm1d(8), nn.ReLU(), nn.Linear(8, 4, bias=False), nn.BatchNorm1d(4))
    common.init_by_name(m)
    return m


def seeded_grads(m, it, big):
    for j, (n, p) in enumerate(m.named_parameters()):
        u = torch.from_numpy(common._hash01(p.numel(), 100 * it + j)).reshape(p.shape)
        p.grad = (u - 0.5) * (40.0 if big else 0.5)


@pytest.mark.parametrize("tag,lr,wd", [("det", 0.01, 0.01), ("occ", 0.003, 0.001)])
def test_group_optimizer_vs_reference(tag, lr, wd):
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optim.npz"))
    per_epoch, epochs = [int(v) for v in g["meta"]]
    m = tiny()
    opt = GroupOptimizer([dict(params=list(m.parameters()), lr=lr, weight_decay=wd, grad_norm_clip=10.0, moms=(0.95, 0.85), div_factor=10,
                               pct_start=0.4, lr_clip=1e-7)], total_steps=per_epoch * epochs)
    snaps = {int(i): k for k, i in enumerate(g["snap_iters"])}
    for it in range(per_epoch * epochs + 3):
        assert opt.groups[0]["lr"] == pytest.approx(float(g[tag + "_lr"][it]), rel=1e-12, abs=0)
        assert opt.groups[0]["mom"] == pytest.approx(float(g[tag + "_mom"][it]), rel=1e-12, abs=0)
        opt.zero_grad()
        seeded_grads(m, it, big=(it % 3 == 0))
        opt.step()
        if it in snaps:
            got = np.concatenate([p.detach().numpy().reshape(-1) for p in m.parameters()])
            np.testing.assert_allclose(got, g[tag + "_params"][snaps[it]], rtol=2e-5, atol=2e-7, err_msg="after step %d" % it)


def test_parameters_without_gradient_are_skipped_like_torch_adam():
    m = tiny()
    ps = list(m.parameters())
    opt = GroupOptimizer([dict(params=ps, lr=0.01, weight_decay=0.01, grad_norm_clip=10.0)], total_steps=100)
    before = [p.detach().clone() for p in ps]
    seeded_grads(m, 0, False)
    ps[2].grad = None
    opt.step()
    assert torch.equal(ps[2], before[2]) and float(opt.groups[0]["steps"][2]) == 0.0      # untouched: no decay, no moment update
    assert not torch.equal(ps[0], before[0]) and float(opt.groups[0]["steps"][0]) == 1.0


def test_one_cycle_endpoints():
    s = OneCycle(100, 0.01, (0.95, 0.85), 10, 0.4)
    assert s.initial() == (0.001, 0.95)
    assert s.at(40)[0] == pytest.approx(0.01) and s.at(40)[1] == pytest.approx(0.85)
    assert s.at(100)[0] == pytest.approx(0.001 / 1e4) and s.at(1000) == s.at(100)


def test_groups_stepped_separately_equal_one_step():
    """bench.py steps the detection group while the occupancy branch is still in backward: group by group == all at once,
    schedules included"""
    torch.manual_seed(3)

    def make():
        torch.manual_seed(4)
        a, b = torch.nn.Linear(5, 7), torch.nn.Linear(7, 3)
        opt = GroupOptimizer([dict(params=list(a.parameters()), lr=0.003, weight_decay=0.001, grad_norm_clip=0.5),
                              dict(params=list(b.parameters()), lr=0.01, weight_decay=0.01, grad_norm_clip=0.5)], total_steps=20)
        return a, b, opt

    runs = []
    for split in (False, True):
        a, b, opt = make()
        torch.manual_seed(5)
        for it in range(6):
            x = torch.randn(4, 5)
            opt.zero_grad()
            b(a(x)).square().sum().backward()
            if split:
                opt.step(groups=[1])
                assert opt.iteration == it + 1
                opt.step(groups=[0])
            else:
                opt.step()
            assert opt.iteration == it + 1
        runs.append([p.detach().clone() for p in list(a.parameters()) + list(b.parameters())] + [torch.tensor(opt.lrs())])
    for u, v in zip(*runs):
        assert torch.equal(u, v)


def test_chunk_tables_cover_every_parameter_once():
    """chunk tables of csrc/optim.hip (shared by the flat optimizer step and the one-launch bucket pack): chunks of <= 1024 elements,
    none straddling two parameters, each element of the flat layout covered exactly once, in order"""
    from btcdet_amd.train_step import chunk_tables
    sizes = [5, 1024, 1025, 1, 3000, 2048] + [7] * 460          # > 448 parameters: the pointer table wraps
    t = chunk_tables(sizes, torch.device("cpu"))
    seg, off, ln, flat, seg0 = (t[k].tolist() if k != "seg0" else t[k] for k in ("seg", "off", "len", "flat", "seg0"))
    assert len(seg0) == len(sizes) + 1 and seg0[0] == 0 and seg0[-1] == len(seg)
    starts = [0]
    for n in sizes:
        starts.append(starts[-1] + n)
    covered = 0
    for i, n in enumerate(sizes):
        chunks = range(seg0[i], seg0[i + 1])
        assert len(chunks) == -(-n // 1024)
        pos = 0
        for c in chunks:
            assert seg[c] == i % 448 and off[c] == pos and 1 <= ln[c] <= 1024 and flat[c] == starts[i] + pos
            pos += ln[c]
        assert pos == n
        covered += pos
    assert covered == sum(sizes)
