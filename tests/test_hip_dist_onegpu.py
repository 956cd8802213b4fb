"""BASELINE.json configs[3] as far as a 1-GPU box can take it: the N > 1 data-parallel step with the REAL hot path -- BtcHotPath +
btcdet_amd.trainer.HotPathTrainer + BucketedGradSync + GroupOptimizer reading the reduced gradients from the flat buckets -- as TWO
ranks sharing one GPU over gloo (the reducer's host-staged transport), disjoint scenes per rank, the gradient-norm clip biting, three
optimizer steps.  tests/dist_worker.py holds the per-rank body and what it checks; tests/test_hip_rccl.py runs the same body over RCCL
(one rank per GPU)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dist_worker  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("schedule", ["split", "pipelined"])
def test_two_ranks_on_one_gpu_real_hot_path_clip_biting(schedule):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(dist_worker.run, args=(world, 29781 + (schedule == "pipelined"), out, "gloo", schedule, None, 3, True), nprocs=world, join=True)
    bad = {k: v for k, v in out.items() if str(k).startswith("bad_")}
    assert not bad, bad     # parameters with a non-finite local gradient (rank, step)
    for r in range(world):
        o = out[r]
        assert o["backend"] == "gloo" and o["world"] == 2 and o["transport"] == "host" and o["it"] == 3
        assert o["pipelined"] == (schedule == "pipelined")
        assert all(n > 4 * dist_worker.CLIP for step in o["norms"] for n in step), o["norms"]     # the clip bites in every step, both groups
        assert o["grad_err"] < 1e-6, o["grad_err"]      # reduced gradient in the buckets == mean of the local gradients
        assert o["param_err"] < 5e-6, o["param_err"]    # parameters == single-process optimizer replay on that mean
    assert abs(out[0]["digest"] - out[1]["digest"]) <= 1e-9 * abs(out[0]["digest"])   # ranks in lockstep after 3 steps
