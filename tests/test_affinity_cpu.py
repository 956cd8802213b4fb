"""NUMA placement helper (btcdet_amd/affinity.py): cpulist parsing and the per-rank slices; no GPU needed"""
import os

from btcdet_amd import affinity


def _restore(avail):
    """every thread of the process back on the whole mask (a thread may end between the listing and the call)"""
    for t in os.listdir("/proc/self/task"):
        try:
            os.sched_setaffinity(int(t), avail)
        except OSError:
            pass


def test_parse_cpulist():
    assert affinity._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert affinity._parse_cpulist("") == []


def test_ranks_take_disjoint_slices(monkeypatch):
    avail = sorted(os.sched_getaffinity(0))
    monkeypatch.setattr(affinity, "local_cpus", lambda d: list(avail))
    got = []
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: got.append(list(cpus)) if pid == 0 else None)   # (pid != 0: the threads that exist already)
    monkeypatch.setattr(affinity, "_one_per_core", lambda cpus: cpus)
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


def test_existing_threads_are_moved_too(monkeypatch):
    """sched_setaffinity(0, ...) moves the calling thread only: the threads the process already has (the HIP runtime's) get the mask one by one"""
    import threading
    avail = sorted(os.sched_getaffinity(0))
    stop = threading.Event()
    tid = []
    th = threading.Thread(target=lambda: (tid.append(threading.get_native_id()), stop.wait(30)))
    th.start()
    try:
        while not tid:
            pass
        monkeypatch.setattr(affinity, "local_cpus", lambda d: list(avail))
        monkeypatch.setattr(affinity, "_one_per_core", lambda cpus: cpus)
        mine = affinity.pin_to_gpu(0, cpus_per_rank=1)
        assert mine == avail[:1]
        assert sorted(os.sched_getaffinity(tid[0])) == mine
    finally:
        stop.set()
        th.join()
        _restore(avail)


def test_one_cpu_per_core():
    cpus = sorted(os.sched_getaffinity(0))
    keep = affinity._one_per_core(cpus)
    assert keep and set(keep) <= set(cpus) and keep == sorted(keep)
    assert affinity._one_per_core(keep) == keep


def test_busy_threads_get_a_cpu_each(monkeypatch):
    """place_thread: the schedule's four busy host threads on the first four CPUs of the process's mask, place_other_threads: every other
    thread on the remaining ones (a thread that owns a CPU keeps it); nothing moves in a process that was not pinned"""
    import threading
    avail = sorted(os.sched_getaffinity(0))
    if len(avail) < 8:
        import pytest
        pytest.skip("needs 8 CPUs")
    monkeypatch.setattr(affinity, "_MINE", [])
    assert affinity.place_thread("train") is False and affinity.place_other_threads() == 0
    assert sorted(os.sched_getaffinity(0)) == avail
    mine = avail[:8]
    monkeypatch.setattr(affinity, "_MINE", list(mine))
    # (a mask of 8 CPUs: one per role, four for the other threads -- affinity._role_width)
    stop, tids = threading.Event(), {}

    def worker(role):
        if role is not None:
            affinity.place_thread(role)
        tids[role] = threading.get_native_id()
        stop.wait(30)
    threads = [threading.Thread(target=worker, args=(r,)) for r in ("occupancy", "prepare", None)]
    for t in threads:
        t.start()
    try:
        while len(tids) < 3:
            pass
        assert affinity.place_thread("train")
        assert affinity.place_other_threads() >= 1
        w = affinity._role_width()
        assert w == 1
        assert sorted(os.sched_getaffinity(0)) == mine[0:w]
        assert sorted(os.sched_getaffinity(tids["occupancy"])) == mine[2 * w:3 * w] and sorted(os.sched_getaffinity(tids["prepare"])) == mine[3 * w:4 * w]
        assert sorted(os.sched_getaffinity(tids[None])) == mine[4 * w:]
    finally:
        stop.set()
        for t in threads:
            t.join()
        _restore(avail)


def test_ranks_share_only_their_own_sockets_cpus(monkeypatch):
    """8 ranks, one per GPU, 4 GPUs per socket: a rank's slice comes out of its socket's list divided by FOUR, and the slices of one
    socket's ranks are disjoint"""
    import torch
    sockets = {0: list(range(0, 64)), 1: list(range(64, 128))}
    monkeypatch.setattr(affinity, "local_cpus", lambda d: list(sockets[d // 4]))
    monkeypatch.setattr(affinity, "_one_per_core", lambda cpus: cpus)
    monkeypatch.setattr(affinity, "_pin_existing_threads", lambda cpus: None)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(128)))
    got = []
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: got.append(list(cpus)))
    for r in range(8):
        mine = affinity.pin_to_gpu(r, local_rank=r, ranks_on_node=8)
        assert len(mine) == 16 and set(mine) <= set(sockets[r // 4])
    assert len({c for m in got for c in m}) == 128


def test_a_thread_started_by_a_placed_thread_does_not_keep_its_cpus(monkeypatch):
    """a helper thread created from a placed thread inherits that thread's mask; place_other_threads moves it to the shared CPUs all the same"""
    import threading
    avail = sorted(os.sched_getaffinity(0))
    if len(avail) < 8:
        import pytest
        pytest.skip("needs 8 CPUs")
    mine = avail[:8]
    monkeypatch.setattr(affinity, "_MINE", list(mine))
    monkeypatch.setattr(affinity, "_PLACED", {})
    stop, tids = threading.Event(), {}

    def child():
        tids["child"] = threading.get_native_id()
        stop.wait(30)

    def parent():
        affinity.place_thread("occupancy")
        tids["parent"] = threading.get_native_id()
        c = threading.Thread(target=child)
        c.start()
        stop.wait(30)
        c.join()
    p = threading.Thread(target=parent)
    p.start()
    try:
        while len(tids) < 2:
            pass
        w = affinity._role_width()
        assert sorted(os.sched_getaffinity(tids["child"])) == mine[2 * w:3 * w]        # inherited
        affinity.place_thread("train")
        affinity.place_other_threads()
        assert sorted(os.sched_getaffinity(tids["parent"])) == mine[2 * w:3 * w]
        assert sorted(os.sched_getaffinity(tids["child"])) == mine[4 * w:]
    finally:
        stop.set()
        p.join()
        _restore(avail)
