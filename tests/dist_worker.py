"""Worker of the multi-process tests of the data-parallel step with the REAL hot path (tests/test_hip_dist_onegpu.py: two ranks
sharing one GPU over gloo; tests/test_hip_rccl.py: one rank per GPU over RCCL).  Every rank runs btcdet_amd.trainer.HotPathTrainer on
disjoint scenes for a few optimizer steps with the gradient-norm clip BITING and checks, step by step:
  * the reduced gradient in the reducer's flat buckets == the mean of the ranks' local gradients (computed independently),
  * the parameters == a single-process GroupOptimizer replay on that mean (norm clip over the bucket views included),
and reports a digest of its parameters so that the caller can assert the ranks stayed in lockstep."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CLIP = 5e-5     # far below the gradient norms of this step (asserted: the detection group's stand-in loss gives ~5e-4): the clip coefficient really rescales the update


def run(rank, world, port, out, backend, schedule, transport=None, steps=3, share_gpu=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if transport:
        os.environ["BTC_SYNC_TRANSPORT"] = transport
    dev_index = 0 if share_gpu else rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        import bench
        from btcdet_amd.btc_path import BtcHotPath
        from btcdet_amd.config import load_cfg
        from btcdet_amd.train_step import GroupOptimizer
        from btcdet_amd.trainer import HotPathTrainer
        torch.manual_seed(666)
        np.random.seed(666)
        model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
        occ = [p for p in model.occ_modules.parameters() if p.requires_grad]
        det = [p for p in model.det_modules.parameters() if p.requires_grad]
        kw = dict(grad_norm_clip=CLIP, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4, lr_clip=1e-7)
        groups = [dict(params=occ, lr=0.003, weight_decay=0.001, **kw), dict(params=det, lr=0.01, weight_decay=0.01, **kw)]
        tr = HotPathTrainer(model, groups=groups, total_steps=1000, schedule=schedule, distributed=True)
        sync = tr.grad_sync
        params = occ + det
        # single-process replay on copies of the (broadcast) start
        ref = [p.detach().clone().requires_grad_(True) for p in params]
        rgroups = [dict(params=ref[:len(occ)], lr=0.003, weight_decay=0.001, **kw), dict(params=ref[len(occ):], lr=0.01, weight_decay=0.01, **kw)]
        ropt = GroupOptimizer(rgroups, total_steps=1000, flat=False)
        batches = bench.build_batches(steps + 1, rank, dev, 2, "kitti")        # disjoint scenes per rank (DistributedSampler shard)
        worst_g = worst_p = 0.0
        norms = []
        for i in range(steps):
            tr.step(batches[i], batches[i + 1])
            torch.cuda.synchronize()
            local = [p.grad.detach().clone() for p in params]              # assign_grads=False: param.grad is the LOCAL gradient
            bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
            if bad:
                good = [n for n, p in model.named_parameters() if p.grad is not None and n.startswith("det_") and torch.isfinite(p.grad).all()]
                out["bad_%d_%d" % (rank, i)] = {"bad": bad, "finite_det": good}
                if os.environ.get("BTC_NAN_DUMP"):
                    import json
                    with open(os.environ["BTC_NAN_DUMP"], "a") as fh:
                        fh.write(json.dumps({"rank": rank, "step": i, "bad": bad, "finite_det": good}) + "\n")
            reduced = [sync.view_of(p).detach().clone() for p in params]
            mean = []
            for g in local:                                                # the mean over ranks, independently of the reducer
                if backend == "nccl":
                    t = g.clone()
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                else:
                    t = g.cpu()
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                    t = t.to(dev)
                mean.append(t / world)
            worst_g = max(worst_g, max(float((a - b).abs().max() / (b.abs().max() + 1e-12)) for a, b in zip(reduced, mean)))
            for r, m in zip(ref, mean):
                r.grad = m
            norms.append([float(torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(m) for m in part])))
                          for part in (mean[:len(occ)], mean[len(occ):])])
            ropt.step()
            worst_p = max(worst_p, max(float((a.detach() - b.detach()).abs().max()) for a, b in zip(params, ref)))
        digest = float(sum(p.detach().double().sum() for p in params))
        out[rank] = dict(grad_err=worst_g, param_err=worst_p, digest=digest, backend=dist.get_backend(), world=dist.get_world_size(),
                         transport=sync.transport, norms=norms, pipelined=bool(tr._step.pipelined), it=tr.optimizer.iteration)
        dist.barrier()
    finally:
        dist.destroy_process_group()
