"""NUMA placement helper (btcdet_amd/affinity.py): cpulist parsing and the per-rank slices; no GPU needed"""
import os

from btcdet_amd import affinity


def test_parse_cpulist():
    assert affinity._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert affinity._parse_cpulist("") == []


def test_ranks_take_disjoint_slices(monkeypatch):
    avail = sorted(os.sched_getaffinity(0))
    monkeypatch.setattr(affinity, "local_cpus", lambda d: list(avail))
    got = []
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: got.append(list(cpus)))
    n = min(2, len(avail))
    for r in range(n):
        mine = affinity.pin_to_gpu(0, local_rank=r, ranks_on_node=n, cpus_per_rank=max(1, len(avail) // n))
        assert mine and set(mine) <= set(avail)
    assert len(got) == n and (n < 2 or not (set(got[0]) & set(got[1])))


def test_no_topology_means_no_pinning(monkeypatch):
    monkeypatch.setattr(affinity, "local_cpus", lambda d: None)
    assert affinity.pin_to_gpu(0) is None
    monkeypatch.setenv("BTC_PIN_CPUS", "0")
    assert affinity.pin_to_gpu(0) is None
