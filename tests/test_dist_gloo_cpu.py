"""world_size-2 gloo run of the data-parallel scaffolding bench.py uses (one process per device, DDP gradient
all-reduce, barrier-bracketed timing with max over ranks, disjoint scene shards).  The HIP kernels cannot run on CPU,
so the model here is a small torch module; what is covered is the N > 1 control path."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(666)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU(), torch.nn.Linear(16, 1))
    ddp = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=False)
    seeds = bench.rank_seeds(rank, 0)
    x = torch.from_numpy(np.random.default_rng(seeds[0]).standard_normal((32, 8)).astype(np.float32))
    dist.barrier()
    loss = ddp(x).pow(2).mean()
    loss.backward()
    dist.barrier()
    dt = bench.max_over_ranks(0.1 * (rank + 1), dist, torch.device("cpu"))
    g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    local = torch.autograd.grad(model(x).pow(2).mean(), list(model.parameters()))
    out[rank] = (g.numpy(), torch.cat([t.reshape(-1) for t in local]).numpy(), dt, seeds)
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_gloo_world2():
    world, port = 2, 29731
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    g0, l0, dt0, s0 = out[0]
    g1, l1, dt1, s1 = out[1]
    np.testing.assert_allclose(g0, g1, rtol=0, atol=0)                      # all ranks hold the same reduced gradient
    np.testing.assert_allclose(g0, 0.5 * (l0 + l1), rtol=1e-5, atol=1e-6)  # = mean of the per-rank gradients
    assert dt0 == dt1 == 0.2                                                # max over ranks
    assert not set(s0) & set(s1)                                            # disjoint scene shards


def _sync_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from btcdet_amd.grad_sync import BucketedGradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1)
    # two detached branches like the hot path: `a` (created first, "occupancy") and `b` (created later, "detection")
    a = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 4))
    b = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.ReLU(), torch.nn.Linear(8, 2))
    launched = []
    sync = BucketedGradSync([(list(b.parameters()), a[2].weight), (list(a.parameters()), None)])
    orig = sync.launch
    sync.launch = lambda bk, only_if_complete=False: (launched.append((sync.buckets.index(bk), only_if_complete,
                                                                         all(p.grad is not None for p in bk.params))), orig(bk, only_if_complete))[1]
    for h in sync._handles:
        h.remove()
    sync._handles = [a[2].weight.register_post_accumulate_grad_hook(lambda p: sync.launch(sync.buckets[0], only_if_complete=True))]
    x = torch.from_numpy(np.random.default_rng(10 + rank).standard_normal((16, 6)).astype(np.float32))
    local = None
    for it in range(2):
        for p in list(a.parameters()) + list(b.parameters()):
            p.grad = None
        ya = a(x)
        loss = ya.pow(2).mean() + b(ya.detach()).pow(2).mean()
        local = torch.autograd.grad(loss, list(b.parameters()) + list(a.parameters()), retain_graph=True)
        loss.backward()
        sync.finish()
    g = torch.cat([p.grad.reshape(-1) for p in list(b.parameters()) + list(a.parameters())])
    out[rank] = (g.numpy().copy(), torch.cat([t.reshape(-1) for t in local]).numpy(), launched)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_grad_sync_gloo_world2():
    """btcdet_amd/grad_sync.py: the later-created (detection-like) branch's bucket is launched from the hook on the first
    parameter of the other branch that receives a gradient -- with all of its gradients present -- and every rank ends with the
    mean gradient in param.grad"""
    world, port = 2, 29741
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sync_worker, args=(world, port, out), nprocs=world, join=True)
    g0, l0, launched0 = out[0]
    g1, l1, _ = out[1]
    np.testing.assert_allclose(g0, g1, rtol=0, atol=0)
    np.testing.assert_allclose(g0, 0.5 * (l0 + l1), rtol=1e-5, atol=1e-7)
    early = [e for e in launched0 if e[0] == 0 and e[1]]
    assert early and all(e[2] for e in early)   # the early launch of bucket 0 saw a complete bucket


def _one_bucket_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import bench
    from btcdet_amd.grad_sync import BucketedGradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(2)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    ref = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    ref.load_state_dict(net.state_dict())
    params = list(net.parameters())
    sync = BucketedGradSync([(params, None)], assign_grads=False)     # bench.py's configuration for N > 1
    try:
        from btcdet_amd.train_step import GroupOptimizer
        opt = GroupOptimizer([{"params": params, "lr": 1e-2, "weight_decay": 0.01, "grad_norm_clip": 10.0}], total_steps=50)
        opt.read_grads_from(sync.view_of, sync.has_grad)
        ropt = GroupOptimizer([{"params": list(ref.parameters()), "lr": 1e-2, "weight_decay": 0.01, "grad_norm_clip": 10.0}], total_steps=50)
        for it in range(3):
            xs = [torch.from_numpy(np.random.default_rng(100 * it + r).standard_normal((8, 5)).astype(np.float32)) for r in range(world)]
            opt.zero_grad()
            net(xs[rank]).pow(2).mean().backward()
            local = [p.grad.clone() for p in params]
            sync.launch_ready()                                        # everything is present: the bucket goes now
            sync.finish()
            assert all(torch.equal(p.grad, g) for p, g in zip(params, local))   # param.grad untouched (assign_grads=False)
            opt.step()
            ropt.zero_grad()
            (sum(ref(x).pow(2).mean() for x in xs) / world).backward()          # the mean over ranks, computed locally
            ropt.step()
        out[rank] = ([p.detach().numpy().copy() for p in params], [p.detach().numpy().copy() for p in ref.parameters()])
    except (RuntimeError, NotImplementedError) as e:
        out[rank] = "skip: %s" % e
    dist.barrier()
    dist.destroy_process_group()


def test_one_bucket_reducer_feeds_the_optimizer_gloo_world2():
    """bench.py's N > 1 configuration: one flat bucket sent after backward, the optimizer (norm clip + decoupled decay + fused
    Adam, btcdet_amd.train_step) reads the reduced gradients from the bucket's slices (param.grad is left alone); the
    parameters follow a single-process run of the same optimizer on the mean loss"""
    world, port = 2, 29751
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_one_bucket_worker, args=(world, port, out), nprocs=world, join=True)
    if isinstance(out[0], str):
        import pytest
        pytest.skip(out[0])
    (p0, r0), (p1, _) = out[0], out[1]
    for a, b, r in zip(p0, p1, r0):
        np.testing.assert_allclose(a, b, rtol=0, atol=0)              # ranks stay in lockstep
        np.testing.assert_allclose(a, r, rtol=2e-5, atol=2e-6)


def _missing_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from btcdet_amd.grad_sync import BucketedGradSync, GradSyncError
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(3)
    net = torch.nn.ModuleList([torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)])
    params = list(net.parameters())
    sync = BucketedGradSync([(params, None)], assign_grads=False)
    x = torch.ones(2, 4)
    events = []
    for it in range(3):
        for p in params:
            p.grad = None
        # step 0: every rank uses both layers; step 1: rank 1 skips the second layer (its parameters get no gradient there);
        # step 2: consistent again -- the launch of step 2 looks at step 1's reduced count and must raise on BOTH ranks
        y = net[0](x)
        if not (it == 1 and rank == 1):
            y = net[1](y)
        y.sum().backward()
        try:
            sync.finish()
            events.append(("ok", sorted(sync.local_missing()) != [], all(sync.has_grad(p) for p in params)))
        except GradSyncError as e:
            events.append(("raised", str(e)[:40]))
            break
    out[rank] = events
    dist.barrier()
    dist.destroy_process_group()


def test_reducer_raises_on_every_rank_when_ranks_disagree_on_used_parameters_gloo_world2():
    """a parameter with a gradient on one rank and none on another used to make the ranks apply
    different updates silently.  Now every parameter of a reduced bucket counts as present on every rank (zeros from the rank
    that had none), and the disagreement -- carried through the same collective -- raises on every rank one step later, the
    contract of DistributedDataParallel(find_unused_parameters=False) the reference trains under (tools/train.py:166-168)."""
    world, port = 2, 29771
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_missing_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        ev = out[r]
        assert ev[0][0] == "ok" and ev[0][1] is False and ev[0][2] is True
        assert ev[1][0] == "ok" and ev[1][2] is True            # the inconsistent step itself: same (mean) update everywhere
        assert ev[1][1] is (r == 1)                             # only rank 1 had local gaps
        assert ev[2][0] == "raised", ev                         # ... and both ranks raise at the next launch


def _convert_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from btcdet_amd.spconv import fused_bn
    from btcdet_amd.spconv.modules import SparseSequential
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = torch.nn.Module()
        net.sparse = SparseSequential(torch.nn.BatchNorm1d(8), torch.nn.ReLU())          # routed through this package's kernels: marked
        net.bev = torch.nn.Sequential(torch.nn.Conv2d(4, 4, 1), torch.nn.BatchNorm2d(4))   # BaseBEVBackbone-style: torch's SyncBatchNorm
        net.mlp = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.BatchNorm1d(8))      # a plain BatchNorm1d outside SparseSequential
        keys = list(net.state_dict().keys())
        w2d = net.bev[1].weight
        n = fused_bn.convert_sync_batchnorm(net)
        empty = fused_bn.combine_stats(torch.zeros((2, 2 * 8 + 1)), 8)
        out[rank] = dict(n=n, sparse_marked=type(net.sparse[0]) is torch.nn.BatchNorm1d and net.sparse[0].sync_group is not None,
                         bev=type(net.bev[1]).__name__, mlp=type(net.mlp[1]).__name__, same_param=net.bev[1].weight is w2d,
                         keys_same=list(net.state_dict().keys()) == keys, again=fused_bn.convert_sync_batchnorm(net),
                         empty_finite=bool(all(torch.isfinite(t).all() for t in empty)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_convert_sync_batchnorm_covers_every_batchnorm():
    """the reference's torch.nn.SyncBatchNorm.convert_sync_batchnorm(model) converts EVERY _BatchNorm; so does this one
    (sparse BatchNorm1d layers marked, everything else replaced by torch's SyncBatchNorm), and counts only what synchronises"""
    world, port = 2, 29741
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_convert_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        o = out[r]
        assert o["n"] == 3 and o["sparse_marked"] and o["bev"] == "SyncBatchNorm" and o["mlp"] == "SyncBatchNorm"
        assert o["same_param"] and o["keys_same"] and o["empty_finite"]
        assert o["again"] == 1      # idempotent: only the marked layer is counted again, nothing is replaced twice
