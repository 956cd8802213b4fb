"""The data-parallel step over RCCL (backend "nccl" on ROCm) with the REAL hot path (BtcHotPath + bench.make_step + the bucketed
gradient reducer): every rank's parameters after a step equal a single-process step on the mean of the per-rank gradients, the
reduced gradient in the flat buckets equals the mean of the ranks' local gradients, and the ranks stay in lockstep.

world size 2 needs two GPUs (skipped on a 1-GPU box; the driver's multi-GPU node runs it); world size 1 exercises the same code
path -- process group on RCCL, ncclAvg all-reduce of both buckets, the split backward that overlaps the detection bucket with the
occupancy branch's backward -- on one GPU."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import bench
    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    from btcdet_amd.grad_sync import BucketedGradSync
    from btcdet_amd.train_step import GroupOptimizer
    torch.manual_seed(666)
    np.random.seed(666)
    model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
    for p in model.parameters():
        dist.broadcast(p.data, src=0)
    occ = [p for p in model.occ_modules.parameters() if p.requires_grad]
    det = [p for p in model.det_modules.parameters() if p.requires_grad]
    sync = BucketedGradSync([(det, None), (occ, None)], assign_grads=False)
    sync.split_backward = True
    opt = GroupOptimizer([dict(params=occ, lr=0.003, weight_decay=0.001, grad_norm_clip=1e9),
                          dict(params=det, lr=0.01, weight_decay=0.01, grad_norm_clip=1e9)], total_steps=1000)
    # (norm clip effectively off: the clip coefficient rides into the fused Adam kernel as grad_scale, which rescales the
    # gradients -- here the buckets -- in place; the clip itself is pinned to the reference by tests/test_train_step_cpu.py)
    opt.read_grads_from(sync.view_of, sync.has_grad)
    before = [p.detach().clone() for p in det + occ]
    batch = bench.build_batches(1, rank, dev, 2, "kitti")[0]        # disjoint scenes per rank (DistributedSampler shard)
    step = bench.make_step(model, model, model.dataset.data_processor, [opt], sync)
    step(batch)
    torch.cuda.synchronize()
    local = [p.grad.detach().clone() for p in det + occ]            # assign_grads=False: param.grad is the LOCAL gradient
    reduced = [sync.view_of(p).detach().clone() for p in det + occ]
    # mean of the ranks' local gradients, computed independently of the reducer
    mean = []
    for g in local:
        t = g.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        mean.append(t / world)
    err = max(float((a - b).abs().max() / (b.abs().max() + 1e-12)) for a, b in zip(reduced, mean))
    # single-process replay of the optimizer on the mean gradients
    ref = [b.clone().requires_grad_(True) for b in before]
    ropt = GroupOptimizer([dict(params=ref[len(det):], lr=0.003, weight_decay=0.001, grad_norm_clip=1e9),
                           dict(params=ref[:len(det)], lr=0.01, weight_decay=0.01, grad_norm_clip=1e9)], total_steps=1000)
    for r, m in zip(ref, mean):
        r.grad = m
    ropt.step()
    perr = max(float((a.detach() - b.detach()).abs().max()) for a, b in zip(det + occ, ref))
    digest = float(sum(p.detach().double().sum() for p in det + occ))
    out[rank] = (err, perr, digest, dist.get_backend(), dist.get_world_size())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_hot_path_step_over_rccl(world):
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs" % world)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, 29761 + world, out), nprocs=world, join=True)
    for r in range(world):
        err, perr, digest, backend, ws = out[r]
        assert backend == "nccl" and ws == world
        assert err < 1e-6, err            # reduced gradient in the buckets == mean of the local gradients (ncclAvg)
        assert perr < 1e-6, perr          # parameters == single-process optimizer step on that mean
    assert len({round(out[r][2], 6) for r in range(world)}) == 1      # ranks in lockstep
