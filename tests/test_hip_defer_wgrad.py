"""Deferred weight-gradient join (btcdet_amd/csrc/binding.cpp conv_bwd, ops.set_defer_wgrad_join): every wgrad of the backward
pass runs on the side stream and is joined once by an autograd-engine callback.  The gradients must be the ones the
in-order schedule produces on every step of a loop without device synchronisation in between (the rulebooks,
activations and gradients the side stream reads are released by the main stream while the side stream still owes work)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(defer, steps, accumulate=False):
    import bench
    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    from btcdet_amd.spconv import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
    assert ops.set_defer_wgrad_join(defer), "compiled binding missing"
    try:
        batches = bench.build_batches(2, 0, dev)
        proc = model.dataset.data_processor
        params = [p for p in model.parameters() if p.requires_grad]
        out = []
        for it in range(steps):
            b = batches[it % 2]
            bd = proc.forward_batch(b["points"], b["pre_rot_points"], b["scene_offsets"], b["rot_z"])
            bd.update({"batch_size": 2, "points": b["points5"], "gt_boxes": b["gt_boxes"], "gt_boxes_num": b["gt_boxes_num"],
                       "box_mirr_flag": b["box_mirr_flag"], "bm_points": b["bm_points"], "rot_z": b["rot_z"], "is_train": True})
            ret, _, _ = model(bd)
            loss = ret["loss_occ"] + 1e-3 * ret["spatial_features"].pow(2).mean() + 1e-3 * ret["x_combine"].float().pow(2).mean()
            if not accumulate:
                for p in params:
                    p.grad = None
            loss.backward()
            # no synchronize: the clone below is ordered after the engine callback's join on the current stream
            out.append([None if p.grad is None else p.grad.clone() for p in params])
        torch.cuda.synchronize()
        return out
    finally:
        ops.set_defer_wgrad_join(False)


@pytest.mark.parametrize("accumulate", [False, True])
def test_deferred_wgrad_join_gradients_equal(accumulate):
    steps = 6
    ref = _run(False, steps, accumulate)
    got = _run(True, steps, accumulate)
    n_cmp = 0
    for it in range(steps):
        for a, b in zip(ref[it], got[it]):
            assert (a is None) == (b is None)
            if a is not None:
                # 1e-5 of the largest magnitude: float atomics in the occupancy targets (see test_hip_prefetch.py)
                assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-30, "step %d" % it
                n_cmp += 1
    assert n_cmp > 50 * steps
