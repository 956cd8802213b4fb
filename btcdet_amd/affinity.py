"""NUMA placement of a rank's host threads next to its GPU.

One process per GPU: the step is a chain of ~430 small launches, and on a two-socket MI355X host the
process that lands on (or migrates to) the socket the GPU is NOT attached to pays for it on every launch -- bench.py measured
324-341 scenes/s unpinned against 343 +- 3 pinned to the GPU's socket on the same box (tools/ab.sh).  pin_to_gpu() restricts
the calling process (and the threads it starts later: the prefetch worker, the autograd thread) to a slice of the CPUs that
sysfs lists as local to the GPU's PCI function.  Nothing happens when the topology cannot be read (containers without sysfs,
non-Linux) or when BTC_PIN_CPUS=0."""
import os


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
    per = max(1, min(cpus_per_rank, len(allowed) // max(1, min(ranks_on_node, len(allowed)))))
    start = (local_rank * per) % len(allowed)
    mine = (allowed + allowed)[start:start + per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return mine
