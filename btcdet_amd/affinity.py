"""NUMA placement of a rank's host threads next to its GPU.

One process per GPU: the step is a chain of ~430 small launches, and on a two-socket MI355X host the
process that lands on (or migrates to) the socket the GPU is NOT attached to pays for it on every launch -- bench.py measured
324-341 scenes/s unpinned against 343 +- 3 pinned to the GPU's socket on the same box (tools/ab.sh).  pin_to_gpu() restricts
the process -- every thread it has and the threads it starts later -- to a slice of the CPUs that sysfs lists as local to the GPU's PCI
function, one hardware thread per core; the schedule (trainer.make_step) then gives each of its busy host threads a CPU of its own
inside that slice (place_thread).  Nothing happens when the topology cannot be read (containers without sysfs, non-Linux) or when
BTC_PIN_CPUS=0."""
import os
import threading


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return cpus


def local_cpus(device_index):
    """CPUs of the NUMA node the GPU `device_index` hangs off (sysfs local_cpulist of its PCI function), or None"""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        with open("/sys/bus/pci/devices/%s/local_cpulist" % bdf) as f:
            cpus = _parse_cpulist(f.read())
        return cpus or None
    except Exception:
        return None


def pin_to_gpu(device_index, local_rank=0, ranks_on_node=1, cpus_per_rank=16):
    """restrict this process to `cpus_per_rank` CPUs local to its GPU; ranks that share a NUMA node take disjoint slices
    (by local rank) as long as the node has enough CPUs.  Returns the CPU list it set, or None."""
    if os.environ.get("BTC_PIN_CPUS", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = local_cpus(device_index)
    if not cpus:
        return None
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
    if not allowed:
        return None
    allowed = _one_per_core(allowed)
    share, slot = _sharers(device_index, cpus, local_rank, ranks_on_node)
    per = max(1, min(cpus_per_rank, len(allowed) // max(1, min(share, len(allowed)))))
    start = (slot * per) % len(allowed)
    mine = (allowed + allowed)[start:start + per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    _pin_existing_threads(mine)
    _MINE[:] = mine
    return mine


def _sharers(device_index, cpus, local_rank, ranks_on_node):
    """(how many ranks divide this GPU's local CPU list, this rank's position among them).  With one rank per visible GPU -- the launch
    the reference and bench.py use -- those are the GPUs whose local list is the same (the GPUs of one socket: 4 of 8 on a two-socket
    host, so a rank gets 16 of the socket's 64 cores, not 8); otherwise every rank of the node is assumed to share it."""
    try:
        import torch
        n_dev = torch.cuda.device_count()
        if n_dev > 1 and ranks_on_node == n_dev:
            same = [d for d in range(n_dev) if local_cpus(d) == cpus]
            if device_index in same:
                return len(same), same.index(device_index)
    except Exception:
        pass
    return ranks_on_node, local_rank


def _one_per_core(cpus):
    """the first hardware thread of every core in `cpus` (sysfs thread_siblings_list); `cpus` unchanged when the topology is unreadable"""
    keep, seen = [], set()
    for c in cpus:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                core = tuple(_parse_cpulist(f.read()))
        except (OSError, ValueError):
            return cpus
        if core not in seen:
            seen.add(core)
            keep.append(c)
    return keep or cpus


def _pin_existing_threads(cpus):
    """sched_setaffinity(0, ...) moves the CALLING thread only (later threads inherit its mask); the threads the process already has --
    the HIP runtime's, started when the device was opened to read its PCI address -- keep theirs.  One of those is busy for the whole
    step (it handles the completion signals of ~330 launches) and floated over all 256 CPUs of the host (tools/step_phases.py).  The mask
    is 16 CPUs of 16 different cores; place_thread / place_other_threads then separate the busy threads inside it."""
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        return
    for tid in tids:
        try:
            os.sched_setaffinity(tid, cpus)
        except OSError:
            pass


_PLACED = {}      # native thread id -> role, of the threads place_thread has placed
_MINE = []        # the CPUs pin_to_gpu gave this process (empty: not pinned -- threads are then left where the scheduler puts them)
ROLES = ("train", "autograd", "occupancy", "prepare")
ROLE_CPUS = 2     # CPUs (of different cores) per role: the thread has somewhere to go when another tenant's thread sits on one of them.  Same
#                   box, six interleaved 80-step runs each: 1 CPU 547 / 548 / 551 / 547 scenes/s, 2 CPUs 553 / 551 / 551 / 551, 3 CPUs (four left
#                   for the runtime's threads) 536 / 536 / 535 / 536; with a neighbour loading the host (load average 9 -> 65) all three lose 5-9 %


def _role_width():
    """ROLE_CPUS while that leaves at least four CPUs of the mask to the other threads, else 1"""
    return ROLE_CPUS if len(_MINE) >= len(ROLES) * ROLE_CPUS + 4 else 1


def place_thread(role, tid=0):
    """give one of the schedule's busy host threads CPUs of its own: the training thread, autograd's device thread, the occupancy
    worker and the prepare worker of trainer.make_step take ROLE_CPUS each from the front of the process's mask (16 CPUs on 16 different
    cores, pin_to_gpu), `place_other_threads` confines everything else to the remaining ones.  Why: the step is as fast as its slowest host
    thread, and a floating thread shares a core now and then -- with another of these four, or with the HIP runtime's signal thread,
    which is busy for the whole step and, woken from the training thread, tends to land on its core's SMT sibling.  Same box, same
    build, 80-step runs (profiles/r06_thread_placement_ab.txt): process-wide mask only 460-557 scenes/s, CPUs per thread 553-560.
    No-op when the process is not pinned (BTC_PIN_CPUS=0, no sysfs topology, fewer than 8 CPUs)."""
    if len(_MINE) < 8:
        return False
    try:
        k, w = ROLES.index(role), _role_width()
        os.sched_setaffinity(tid, set(_MINE[w * k:w * k + w]))
        _PLACED[tid or threading.get_native_id()] = role
        return True
    except OSError:
        return False


def place_other_threads():
    """autograd's device threads (created by the first backward pass; named pt_autograd_<n>) -> their CPU; every thread that has not been
    given a CPU of its own (the HIP runtime's, torch's pools) -> the CPUs no role owns.  Called by the schedule after its first steps."""
    if len(_MINE) < 8:
        return 0
    rest, n = set(_MINE[len(ROLES) * _role_width():]), 0
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        return 0
    for t in tids:
        try:
            with open("/proc/self/task/%d/comm" % t) as f:
                name = f.read().strip()
            if name.startswith("pt_autograd"):
                n += int(place_thread("autograd", t))
            elif t not in _PLACED and set(os.sched_getaffinity(t)) != rest:
                # (by thread id, not by the size of its mask: a thread that one of the placed threads STARTED -- a communication library's
                # helpers at the first collective -- inherits that thread's two CPUs and would sit on them for good)
                os.sched_setaffinity(t, rest)
                n += 1
        except OSError:
            pass
    return n
