"""Canary for BtcHotPath._borrow.  Under the pipelined schedule the occupancy branch's outputs are produced
on one stream and consumed by the detection branch on another; instead of ~40 record_stream calls per step the tensors are HELD until
the consuming stream has passed the end of the step that read them, and dropped when that event has completed.  If that bookkeeping let
go of a tensor too early, its block could be handed out again while the detection branch still reads it -- silently.

Here every tensor is POISONED (NaN / -1, from a third stream that waits for nothing) at the moment _borrow is about to drop it: a reader
that is still in flight, or any later use, turns the step's losses into NaN or moves them away from the in-order schedule's."""
import math
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_borrowed_tensors_are_dead_when_they_are_dropped(monkeypatch):
    import test_hip_prefetch as tp
    from btcdet_amd.btc_path import BtcHotPath
    poison_stream = torch.cuda.Stream()
    poisoned = [0]
    orig = BtcHotPath._borrow

    def canary(self, batch_dict):
        pend = self.__dict__.setdefault("_borrowed", [])
        for k, b in enumerate(pend):
            if len(pend) - k < 3 or b["ended"] is None or not b["ended"].query():   # _borrow drops completed generations, oldest first, down to two
                break
            if True:
                with torch.cuda.stream(poison_stream), torch.no_grad():
                    for i in range(len(b["refs"])):
                        t = b["refs"][i]
                        # only a tensor whose LAST reference is the borrow's: one that autograd saved for the occupancy branch's backward
                        # (which runs on the producer's stream and may lag behind this event) or that another object still holds is
                        # legitimately alive, and its block is not up for reuse either
                        if torch.is_tensor(t) and t.is_cuda and t.numel() and not t._is_view() and t._use_count() == 1 \
                                and sys.getrefcount(t) <= 3 and torch._C._storage_Use_Count(t.untyped_storage()._cdata) <= 2:
                            t.fill_(float("nan") if t.is_floating_point() else (True if t.dtype == torch.bool else -1))
                            poisoned[0] += 1
                        del t
                poison_stream.synchronize()     # the fills must have LANDED before _borrow frees the blocks (the allocator does not know this stream)
        return orig(self, batch_dict)

    steps = 10
    a, na, _ = tp._run_training("plain", steps)
    monkeypatch.setattr(BtcHotPath, "_borrow", canary)
    c, nc, _ = tp._run_training("pipeline", steps)
    torch.cuda.synchronize()
    assert poisoned[0] >= 1, "the canary never fired: no borrowed tensor was released during the run"
    assert all(math.isfinite(x) for x in c) and all(math.isfinite(x) for x in nc), c
    dev = max(abs(x - y) / abs(x) for x, y in zip(a, c))
    print("%d tensors poisoned at their release; plain vs pipelined relative loss deviation %.2e" % (poisoned[0], dev))
    assert dev <= 2e-4, (a, c)
    for x, y in zip(na, nc):
        assert abs(x - y) <= 1e-4 * abs(x)
