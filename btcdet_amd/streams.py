"""Streams whose launches really run side by side.

HIP deals every stream it creates to one of GPU_MAX_HW_QUEUES hardware queues (default 4: the least referenced one at that moment), the
queues sit on the command processor's four pipes, and in-order chains of launches on two streams overlap only when the streams are on
different PIPES (tools/queue_lab.py, fifteen dependent 20 us idle waves per stream, host time for a pair of streams: ~350 us side by side,
~660 us for two streams on one queue, ~850 us for two queues of one pipe; with eight queues the null stream, ten streams of torch's pool
and four of its high-priority pool fall into the classes {null, n3, h0} {n0, n4, n9, h1} {n1, n5, n8, h2} {n2, n6, n7, h3}).  The schedules
of trainer.make_step put the occupancy chain, the detection chain, the next batch's front and the weight gradients on four streams so that
they overlap; which of them shared a pipe used to depend on how many streams the process had created before -- torch's pools, a process
group -- and decided between 3.6 and 5.5 ms per step (tools/queue_probe.py: n unused streams created first, same box: 551 / 386 / 500 / 362 /
552 / 387 / 436 / 466 / 493 scenes/s for n = 0 .. 8; a process group shifted the deal the same way: 556 -> 470 at world size 1).

`same_queue(a, b)` asks the hardware: a chain of idle waves on each of the two streams, host time until both are done -- one chain's time
when they run side by side, two or more when they share a queue or a pipe.  `distinct_stream(others)` draws streams from torch's pool until
one overlaps with all of `others`."""
import time

import torch

from ._lib import check, lib

LINKS, LINK_US = 8, 30


def same_queue(a, b, links=LINKS, link_us=LINK_US):
    """True when in-order chains of launches on streams `a` and `b` do not overlap (one hardware queue, or two queues of one pipe).  Drains the
    device; ~1 ms."""
    if a.cuda_stream == b.cuda_stream:
        return True
    dev = a.device
    best = None
    for _ in range(2):      # (the shorter of two tries: a preempted host thread must not look like a shared queue)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(links):
            check(lib().btc_spin(int(link_us), a.cuda_stream), "btc_spin")
            check(lib().btc_spin(int(link_us), b.cuda_stream), "btc_spin")
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) * 1e6
        best = dt if best is None else min(best, dt)
    return best > 1.5 * links * link_us + 60.0


def distinct_stream(others, device=None, priority=0, tries=24):
    """a stream (of the given priority) whose chains overlap with those of every stream in `others` -> (stream, True), or after `tries` draws
    from torch's pool the last one drawn and False (there are four pipes: a fifth busy stream shares one)"""
    s = None
    for _ in range(tries):
        s = torch.cuda.Stream(device=device, priority=priority)
        if not any(same_queue(o, s) for o in others):
            return s, True
    return s, False
