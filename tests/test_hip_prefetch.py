"""BtcHotPath.prepare: the weight-independent front of a step (both voxelizations, occupancy targets, parameter-free VFE, the
occupancy branch's rulebooks via SparseConvolution.forward_geometry) run ahead of the step that consumes it, on a side stream,
beside the previous step's backward pass (from the training thread, or from a worker thread while the training thread sits in
loss.backward()).  Losses and parameter gradients must equal the in-order schedule's over a loop without device
synchronisation.  The comparison allows 1e-5 of each tensor's largest magnitude: the occupancy targets accumulate per-voxel
residual sums with float atomics (as the reference's scatter-mean does), so two runs of the SAME schedule already differ in the
last bits on some scenes; a race or a recycled buffer shows up as garbage, not as 1e-7."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(mode, steps, defer):
    import bench
    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    from btcdet_amd.spconv import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
    ops.set_defer_wgrad_join(defer)
    try:
        batches = bench.build_batches(3, 0, dev)
        params = [p for p in model.parameters() if p.requires_grad]
        side = torch.cuda.Stream(priority=-1) if mode in ("side_stream", "thread") else None
        pool = None
        if mode == "thread":
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=1)
        out, nxt = [], None
        for it in range(steps):
            b = batches[it % len(batches)]
            if mode == "inline":      # the modules called one after the other by forward(), nothing prepared
                bd = model.assemble(b)
            else:
                bd = nxt if nxt is not None else model.prepare(b)
            assert ("occ_geometry" in bd) == (mode != "inline")
            ret, _, _ = model(bd)
            loss = ret["loss_occ"] + bench.MeanSquare.apply(ret["spatial_features"], 1e-3) + bench.MeanSquare.apply(ret["x_combine"], 1e-3)
            for p in params:
                p.grad = None
            fut = pool.submit(model.prepare, batches[(it + 1) % len(batches)], side) if pool is not None else None
            loss.backward()
            if fut is not None:
                nxt = fut.result()
            elif mode != "inline":
                nxt = model.prepare(batches[(it + 1) % len(batches)], stream=side)
            model.mark_step_end()
            out.append([loss.detach().clone()] + [None if p.grad is None else p.grad.clone() for p in params])
        torch.cuda.synchronize()
        return out
    finally:
        ops.set_defer_wgrad_join(False)


@pytest.mark.parametrize("defer", [False, True])
def test_prepared_steps_equal_inline_steps(defer):
    steps = 7
    ref = _run("inline", steps, defer)
    for mode in ("same_stream", "side_stream", "thread"):
        got = _run(mode, steps, defer)
        n = 0
        for it in range(steps):
            for a, b in zip(ref[it], got[it]):
                assert (a is None) == (b is None)
                if a is not None:
                    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-30, (mode, it)
                    n += 1
        assert n > 50 * steps


def test_forward_geometry_builds_the_rulebooks_forward_uses():
    """after prefetch_geometry the occupancy backbone + head forward must not build a single rulebook"""
    import bench
    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    from btcdet_amd.spconv import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
    b = bench.build_batches(1, 0, dev)[0]
    bd = model.prepare(b)
    coords, geom = bd["occ_geometry"]
    keys_before = set(geom.keys()) | set(geom["__geometry_cache__"].keys())
    built = []
    orig = ops.build_rulebook_g
    ops.build_rulebook_g = lambda *a, **k: (built.append(1), orig(*a, **k))[1]
    try:
        bd["use_occ_prob"] = [True, True]
        for mod in model.occ_module_list[2:4]:
            bd = mod(bd)
    finally:
        ops.build_rulebook_g = orig
    assert not built
    x = bd["encoded_spconv_tensor"]
    assert x.indice_dict is geom
    assert set(geom.keys()) | set(geom["__geometry_cache__"].keys()) == keys_before


def _run_bench_step(mode, steps, defer):
    """bench.make_step itself (no optimizer: gradients are compared, not Adam-normalised updates)"""
    import bench
    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    from btcdet_amd.spconv import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
    ops.set_defer_wgrad_join(defer)
    try:
        batches = bench.build_batches(3, 0, dev)
        params = [p for p in model.parameters() if p.requires_grad]
        side = torch.cuda.Stream(priority=-1) if mode != "plain" else None
        det = torch.cuda.Stream() if mode == "split" else None
        step = bench.make_step(model, model, model.dataset.data_processor, [], None, side, threaded=True, det_stream=det)
        out = []
        for it in range(steps):
            for p in params:
                p.grad = None
            loss = step(batches[it % len(batches)], batches[(it + 1) % len(batches)] if side is not None else None)
            out.append([loss.detach().clone()] + [None if p.grad is None else p.grad.clone() for p in params])
        torch.cuda.synchronize()
        return out
    finally:
        ops.set_defer_wgrad_join(False)


@pytest.mark.parametrize("defer", [False, True])
def test_bench_step_schedules_agree(defer):
    """plain step == step with the next batch prepared from the worker thread == that plus the occupancy branch's backward
    beside the detection branch's forward on a second stream (bench.make_step det_stream)"""
    steps = 7
    ref = _run_bench_step("plain", steps, defer)
    for mode in ("prefetch", "split"):
        got = _run_bench_step(mode, steps, defer)
        n = 0
        for it in range(steps):
            for a, b in zip(ref[it], got[it]):
                assert (a is None) == (b is None)
                if a is not None:
                    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-30, (mode, it)
                    n += 1
        assert n > 50 * steps


def test_chained_layers_equal_layer_by_layer():
    """SparseSequential._chain_plan: with the rulebooks prepared, every stage of the occupancy backbone runs as one compiled call
    (binding.cpp conv_bn_relu_chain); same autograd nodes, so losses and gradients equal the layer-by-layer walk's"""
    from btcdet_amd.spconv import modules
    calls = []
    orig = modules.SparseSequential._run_chain
    modules.SparseSequential._run_chain = lambda self, inp, plan, idx, shp: (calls.append(len(plan)), orig(self, inp, plan, idx, shp))[1]
    try:
        got = _run("same_stream", 4, True)
    finally:
        modules.SparseSequential._run_chain = orig
    assert sum(calls) >= 4 * 12 and max(calls) >= 2     # the 12 conv layers of the occupancy backbone, every step
    modules.CHAIN_LAYERS = False
    try:
        ref = _run("same_stream", 4, True)
    finally:
        modules.CHAIN_LAYERS = True
    for it in range(4):
        for a, b in zip(ref[it], got[it]):
            assert (a is None) == (b is None)
            if a is not None:
                assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-30, it


def _run_training(mode, steps):
    """bench.make_step WITH the reference's optimizer step (two groups): plain one-stream schedule, or the pipelined one (detection
    branch on its own stream, the worker thread steps the occupancy group and runs the next batch's occupancy forward meanwhile)"""
    import bench
    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    from btcdet_amd.spconv import ops
    from btcdet_amd.train_step import GroupOptimizer
    dev = torch.device("cuda:0")
    import numpy as np
    torch.manual_seed(3)
    np.random.seed(3)
    model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
    occ = [p for p in model.occ_modules.parameters() if p.requires_grad]
    det = [p for p in model.det_modules.parameters() if p.requires_grad]
    kw = dict(grad_norm_clip=10.0, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4, lr_clip=1e-7)
    opt = GroupOptimizer([dict(params=occ, lr=0.003, weight_decay=0.001, **kw), dict(params=det, lr=0.01, weight_decay=0.01, **kw)], 1000)
    ops.set_defer_wgrad_join(True)
    try:
        batches = bench.build_batches(3, 0, dev)
        if mode == "plain":
            step = bench.make_step(model, model, model.dataset.data_processor, [opt])
        else:
            step = bench.make_step(model, model, model.dataset.data_processor, [opt], None, torch.cuda.Stream(priority=-1), threaded=True,
                                   det_stream=torch.cuda.Stream())
            assert step.end_stream is not None          # the pipelined variant is the one that runs
        losses = []
        for it in range(steps):
            if mode == "pipeline2":     # the weight-independent front two batches ahead (HotPathTrainer.step's after_next)
                loss = step(batches[it % 3], batches[(it + 1) % 3], batches[(it + 2) % 3])
            else:
                loss = step(batches[it % 3], batches[(it + 1) % 3] if mode != "plain" else None)
            losses.append(loss)
        torch.cuda.synchronize()
        norms = [float(torch.linalg.vector_norm(torch.cat([p.detach().reshape(-1) for p in g]))) for g in (occ, det)]
        return [float(l) for l in losses], norms, opt.iteration
    finally:
        ops.set_defer_wgrad_join(False)


def test_pipelined_training_steps_equal_plain_ones():
    """same losses step by step and same parameters after 8 optimizer steps as the one-stream schedule: every forward pass sees
    exactly the weights it would see there (tolerance: the occupancy targets' float atomics make two plain runs differ too)"""
    steps = 8
    a, na, ia = _run_training("plain", steps)
    b, nb, ib = _run_training("plain", steps)
    noise = max(abs(x - y) / abs(x) for x, y in zip(a, b))
    for mode in ("pipeline", "pipeline2"):
        c, nc, ic = _run_training(mode, steps)
        assert ia == ib == ic == steps
        dev = max(abs(x - y) / abs(x) for x, y in zip(a, c))
        print("relative loss deviation: plain vs plain %.2e, plain vs %s %.2e" % (noise, mode, dev))
        assert dev <= max(20 * noise, 2e-4), (mode, a, c)
        for x, y in zip(na, nc):
            assert abs(x - y) <= 1e-4 * abs(x)


def _run_trainer(steps, env, monkeypatch, overlap_min_rows=None):
    """HotPathTrainer (the object bench.py and a training loop use) for a few optimizer steps under a set of environment switches"""
    import bench
    import numpy as np
    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    from btcdet_amd.spconv import ops
    from btcdet_amd.trainer import HotPathTrainer
    for k in ("BTC_SCHEDULE", "BTC_DEFER_WGRAD", "BTC_TRAINER_TIMING"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    if overlap_min_rows is not None:
        monkeypatch.setattr(ops, "OVERLAP_MIN_ROWS", overlap_min_rows)
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    np.random.seed(3)
    model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
    trainer = HotPathTrainer(model, total_steps=1000, distributed=False)
    try:
        batches = bench.build_batches(3, 0, dev)
        losses = [trainer.step(batches[it % 3], batches[(it + 1) % 3], batches[(it + 2) % 3]) for it in range(steps)]
        torch.cuda.synchronize()
        norms = [float(torch.linalg.vector_norm(torch.cat([p.detach().reshape(-1) for p in g["params"]]))) for g in trainer.optimizer.groups]
        return [float(l) for l in losses], norms, trainer.schedule, ops.defer_wgrad_join_enabled(), dict(getattr(trainer._step, "timing", None) or {})
    finally:
        trainer.finish()
        ops.set_defer_wgrad_join(False)


@pytest.mark.parametrize("env,overlap", [({"BTC_SCHEDULE": "in_order"}, None), ({"BTC_SCHEDULE": "split"}, None), ({"BTC_DEFER_WGRAD": "0"}, None),
                                         ({"BTC_SCHEDULE": "in_order", "BTC_DEFER_WGRAD": "0"}, 2000000000), ({"BTC_DEFER_WGRAD": "0"}, 0),
                                         ({"BTC_TRAINER_TIMING": "1"}, None)],
                         ids=["in_order", "split", "no_deferred_wgrad", "alone", "overlap_every_layer", "host_timing"])
def test_trainer_environment_switches_change_the_schedule_not_the_training(env, overlap, monkeypatch):
    """the switches INTEGRATION.md section 11 lists for HotPathTrainer -- BTC_SCHEDULE (in_order / split / pipelined), BTC_DEFER_WGRAD (weight
    gradients on the side stream with one join per backward), BTC_OVERLAP_MIN_ROWS (ops.OVERLAP_MIN_ROWS: layers from that many rows on
    put their weight gradient beside the next dgrad), BTC_TRAINER_TIMING (host seconds per phase) -- are taken and leave the losses and the
    parameters after 6 optimizer steps where the default schedule puts them (tolerance: the occupancy targets' float atomics)"""
    steps = 6
    a, na, sched_a, defer_a, timing_a = _run_trainer(steps, {}, monkeypatch)
    assert sched_a == "pipelined" and defer_a and not timing_a
    b, nb, sched_b, defer_b, timing_b = _run_trainer(steps, env, monkeypatch, overlap)
    assert sched_b == env.get("BTC_SCHEDULE", "pipelined")
    assert defer_b == (env.get("BTC_DEFER_WGRAD", "1") != "0")
    assert bool(timing_b) == (env.get("BTC_TRAINER_TIMING") == "1")
    dev = max(abs(x - y) / abs(x) for x, y in zip(a, b))
    assert dev <= 5e-4, (env, a, b)
    for x, y in zip(na, nb):
        assert abs(x - y) <= 1e-4 * abs(x)
