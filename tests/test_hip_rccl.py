"""The data-parallel step over RCCL (backend "nccl" on ROCm) with the REAL hot path (BtcHotPath + btcdet_amd.trainer.HotPathTrainer +
the bucketed gradient reducer + the flat optimizer reading the reduced gradients): every rank's parameters after each of three steps
equal a single-process optimizer replay on the mean of the per-rank gradients -- with the gradient-norm clip biting --, the reduced
gradient in the flat buckets equals the mean of the ranks' local gradients, and the ranks stay in lockstep (tests/dist_worker.py).

world size 2 needs two GPUs (skipped on a 1-GPU box; the driver's multi-GPU node runs it); world size 1 exercises the same code
path -- process group on RCCL, the reducer's own RCCL communicator (ncclAllReduce / ncclAvg on the communication stream) or the
process group's all_reduce, both schedules -- on one GPU.  Two ranks on ONE GPU run over gloo in tests/test_hip_dist_onegpu.py."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dist_worker  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,schedule,transport", [(1, "pipelined", "rccl"), (1, "split", "rccl"), (1, "pipelined", "torch"),
                                                      (2, "pipelined", "rccl"), (2, "split", "torch")])
def test_hot_path_step_over_rccl(world, schedule, transport):
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs" % world)
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29761 + world * 10 + (schedule == "split") + 2 * (transport == "torch")
    mp.spawn(dist_worker.run, args=(world, port, out, "nccl", schedule, transport, 3, False), nprocs=world, join=True)
    for r in range(world):
        o = out[r]
        assert o["backend"] == "nccl" and o["world"] == world and o["transport"] == transport and o["it"] == 3
        assert o["pipelined"] == (schedule == "pipelined")
        assert all(n > 4 * dist_worker.CLIP for step in o["norms"] for n in step), o["norms"]     # the clip bites
        assert o["grad_err"] < 1e-6, o["grad_err"]      # reduced gradient in the buckets == mean of the local gradients (ncclAvg)
        assert o["param_err"] < 5e-6, o["param_err"]    # parameters == single-process optimizer replay on that mean
    assert len({round(out[r]["digest"], 6) for r in range(world)}) == 1      # ranks in lockstep
