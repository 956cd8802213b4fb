"""Workers of tests/test_hip_sync_bn.py: two ranks sharing one GPU over gloo."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = (1000, 1733)   # rows per rank: unequal on purpose (the statistics weight ranks by their row counts)
C = 32


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return torch.device("cuda", 0)


def layer(rank, world, port, out):
    """one BatchNorm1d + ReLU over (N_r, C) features with synchronised statistics vs torch on the CONCATENATED batch"""
    dev = _init(rank, world, port)
    try:
        from btcdet_amd.spconv import fused_bn
        g = torch.Generator().manual_seed(5)
        xs = [torch.randn(n, C, generator=g) * 1.7 + 0.3 for n in ROWS]
        dys = [torch.randn(n, C, generator=g) for n in ROWS]
        gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
        # reference: plain torch on the whole batch (CPU, float64)
        ref = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).double()
        ref.weight.data.copy_(gamma); ref.bias.data.copy_(beta)
        xa = torch.cat(xs).double().requires_grad_(True)
        ya = torch.relu(ref(xa))
        ya.backward(torch.cat(dys).double())
        lo, hi = sum(ROWS[:rank]), sum(ROWS[:rank + 1])
        # this rank
        bn = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(dev).train()
        bn.weight.data.copy_(gamma); bn.bias.data.copy_(beta)
        assert fused_bn.convert_sync_batchnorm(bn) == 1 and fused_bn.is_sync(bn) and not fused_bn.fusable(bn)
        x = xs[rank].to(dev).requires_grad_(True)
        y = fused_bn.sync_batch_norm_relu(bn, x, True)
        y.backward(dys[rank].to(dev))
        wg = torch.stack([bn.weight.grad, bn.bias.grad]).cpu()
        dist.all_reduce(wg)    # the parameter gradients are rank-local sums (the gradient reducer averages them): their sum is the batch's
        res = dict(
            y=float((y.detach().cpu().double() - ya.detach()[lo:hi]).abs().max()),
            dx=float((x.grad.cpu().double() - xa.grad[lo:hi]).abs().max()),
            dw=float((wg[0].double() - ref.weight.grad).abs().max() / ref.weight.grad.abs().max()),
            db=float((wg[1].double() - ref.bias.grad).abs().max() / ref.bias.grad.abs().max()),
            rm=float((bn.running_mean.cpu().double() - ref.running_mean).abs().max()),
            rv=float((bn.running_var.cpu().double() - ref.running_var).abs().max()),
            nbt=int(bn.num_batches_tracked))
        bn.eval()
        assert fused_bn.fusable(bn) and not fused_bn.is_sync(bn)     # eval mode: the rank-local fused kernels with the running statistics
        out[rank] = res
        dist.barrier()
    finally:
        dist.destroy_process_group()


def buffers(rank, world, port, out):
    """without sync_bn the running statistics are rank-local; HotPathTrainer.broadcast_buffers() / checkpoint_state() put rank 0's on
    every rank (DistributedDataParallel's broadcast_buffers=True, tools/train.py:166-168)"""
    dev = _init(rank, world, port)
    try:
        import bench
        from btcdet_amd.btc_path import BtcHotPath
        from btcdet_amd.config import load_cfg
        from btcdet_amd.trainer import HotPathTrainer
        torch.manual_seed(666)
        np.random.seed(666)
        net = BtcHotPath(load_cfg(), device=dev).to(dev).train()
        tr = HotPathTrainer(net, distributed=True, schedule="in_order")
        batches = bench.build_batches(2, rank, dev, 2, "kitti")
        tr.step(batches[0], batches[1])
        torch.cuda.synchronize()

        def digest(sd):
            return float(sum(v.double().abs().sum() for k, v in sd.items() if "running_" in k))
        before = digest(net.state_dict())
        state = tr.checkpoint_state(epoch=1)
        after = digest(state["model_state"])
        out[rank] = dict(before=before, after=after, it=state["it"], keys=sorted(state.keys()))
        tr.finish()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def model(rank, world, port, out):
    """HotPathTrainer(sync_bn=True) on the real hot path: two optimizer steps on disjoint scenes; BatchNorm buffers in lockstep"""
    dev = _init(rank, world, port)
    try:
        import bench
        from btcdet_amd.btc_path import BtcHotPath
        from btcdet_amd.config import load_cfg
        from btcdet_amd.trainer import HotPathTrainer
        torch.manual_seed(666)
        np.random.seed(666)
        net = BtcHotPath(load_cfg(), device=dev).to(dev).train()
        tr = HotPathTrainer(net, distributed=True, sync_bn=True)
        batches = bench.build_batches(3, rank, dev, 2, "kitti")
        losses = [float(tr.step(batches[i], batches[i + 1])) for i in range(2)]
        torch.cuda.synchronize()
        bufs = torch.cat([b.detach().double().view(-1) for n, b in net.named_buffers() if "running_" in n]).cpu()
        finite = all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
        out[rank] = dict(n_sync=tr.sync_bn, schedule=tr.schedule, losses=losses, finite=bool(finite), buf_digest=float(bufs.sum()),
                         buf_abs=float(bufs.abs().sum()), params=float(sum(p.detach().double().sum() for p in net.parameters())))
        tr.finish()
        dist.barrier()
    finally:
        dist.destroy_process_group()
