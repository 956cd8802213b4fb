"""BtcHotPath._borrow / mark_step_end (btcdet_amd/btc_path.py): the detection branch holds the tensors the occupancy branch produced on
another stream until the consuming stream has passed the end of the step that read them, instead of registering each with
record_stream.  Host logic only -- events, streams and tensors are stand-ins, no GPU."""
import types

import torch

from btcdet_amd.btc_path import BtcHotPath


class FakeEvent:
    made = []

    def __init__(self):
        self.done = False
        self.recorded_on = None
        FakeEvent.made.append(self)

    def record(self, stream=None):
        self.recorded_on = stream

    def query(self):
        return self.done


class FakeTensor:
    def __init__(self):
        self.recorded = []

    def record_stream(self, s):
        self.recorded.append(s)


def _path(consumed):
    p = types.SimpleNamespace()
    p._prep_state = {"gens": [], "next": consumed + 1, "consumed": consumed}
    return p


def test_borrowed_tensors_are_released_only_after_the_steps_end_event_completed(monkeypatch):
    FakeEvent.made.clear()
    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: "cur")
    p = _path(consumed=7)
    a, b = FakeTensor(), FakeTensor()
    bd = {"__gen_id__": 7, "__produced_here__": [a, b], "x": 1}
    BtcHotPath._borrow(p, bd)
    assert "__produced_here__" not in bd and len(p._borrowed) == 1 and p._borrowed[0]["refs"] == [a, b] and p._borrowed[0]["ended"] is None
    # a step of an EARLIER generation ends: this entry is not touched
    BtcHotPath.mark_step_end(p, stream="det", upto=6)
    assert p._borrowed[0]["ended"] is None and not FakeEvent.made
    # its own step ends: one event, recorded on the consuming stream
    BtcHotPath.mark_step_end(p, stream="det", upto=7)
    ev = p._borrowed[0]["ended"]
    assert ev is FakeEvent.made[0] and ev.recorded_on == "det" and len(FakeEvent.made) == 1
    # the stream has not got there yet: the next step's call keeps the references
    p._prep_state["consumed"] = 8
    BtcHotPath._borrow(p, {"__gen_id__": 8, "__produced_here__": [FakeTensor()]})
    assert [e["id"] for e in p._borrowed] == [7, 8]
    # ... and lets go of a generation once a call three generations later comes in (every forward_det holds a blocking read-back on the
    # consuming stream: by then the event of generation g - 3 has completed -- no driver query); nothing was ever registered with a stream
    BtcHotPath._borrow(p, {"__gen_id__": 9, "__produced_here__": [FakeTensor()]})
    assert [e["id"] for e in p._borrowed] == [7, 8, 9]
    BtcHotPath._borrow(p, {"__gen_id__": 10, "__produced_here__": [FakeTensor()]})
    assert [e["id"] for e in p._borrowed] == [8, 9, 10]     # 7 went (completed); 8 has no end-of-step event yet and stays
    assert a.recorded == [] and b.recorded == []


def test_a_loop_that_never_marks_the_steps_end_falls_back_to_record_stream(monkeypatch):
    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: "cur")
    p = _path(consumed=0)
    ts = [FakeTensor() for _ in range(6)]
    for i, t in enumerate(ts):
        BtcHotPath._borrow(p, {"__gen_id__": i, "__produced_here__": [t]})
    # at most three unmarked generations are held beside the new one; the older ones were registered with the consumer's stream and dropped
    assert [e["id"] for e in p._borrowed] == [2, 3, 4, 5]
    assert [t.recorded for t in ts] == [["cur"], ["cur"], [], [], [], []]
