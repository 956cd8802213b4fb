"""The pipelined schedule (three host threads, four streams, per-stream allocator pools) under a longer run: 48 optimizer steps over 12
distinct batches, parameters and both groups' Adam moments finite at the end.  A non-finite gradient anywhere poisons Adam's moments
for good, so ONE check at the end catches a transient race -- round 3's use-after-free of a temporary weight plane (binding.cpp
conv_bwd q_hold) made one run in three go non-finite within a few steps (tools/nan_stress.py is the open-ended version of this)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pipelined_schedule_stays_finite_over_48_steps():
    import bench
    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    from btcdet_amd.trainer import HotPathTrainer
    dev = torch.device("cuda", 0)
    torch.manual_seed(666)
    np.random.seed(666)
    model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
    tr = HotPathTrainer(model, det_loss=model.det_loss)
    assert tr._step.pipelined
    batches = bench.build_batches(12, 0, dev, 2, "kitti")
    losses = []
    for i in range(48):
        loss = tr.step(batches[i % 12], batches[(i + 1) % 12])
        if i % 8 == 7:
            losses.append(loss)
    torch.cuda.synchronize()
    tr.finish()
    bad = [k for k, p in model.named_parameters() if not torch.isfinite(p).all()]
    assert not bad, bad[:6]
    assert all(bool(torch.isfinite(l).all()) for l in losses)
    for g in tr.optimizer.groups:     # Adam's moments (flat buffers or per-parameter lists)
        moments = list(g["exp_avgs"]) + list(g["exp_avg_sqs"])     # (views into the group's flat moment buffers)
        assert moments and all(bool(torch.isfinite(m).all()) for m in moments)
