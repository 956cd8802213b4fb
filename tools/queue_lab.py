"""Which pairs of streams run concurrently?  For every pair (i, j) of: the current (null) stream, 10 streams of torch's normal pool, 4 of
its high-priority pool: a 300 us idle wave on each, elapsed time from before the first to after both -- ~300 us concurrent, ~600 serialised.
Printed twice (is the relation stable?) as a matrix of 0 (concurrent) / 1 (serialised).  usage: [GPU_MAX_HW_QUEUES=n] python tools/queue_lab.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from btcdet_amd._lib import check, lib  # noqa: E402

torch.cuda.set_device(0)
L = lib()
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(10)] + [torch.cuda.Stream(priority=-1) for _ in range(4)]
names = ["null"] + ["n%d" % i for i in range(10)] + ["h%d" % i for i in range(4)]


CHAIN = int(os.environ.get("CHAIN", "1"))      # CHAIN=k: k dependent idle waves of 300 / k us per stream instead of one (in-order launches of one stream)


def pair(a, b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(CHAIN):
        check(L.btc_spin(300 // CHAIN, a.cuda_stream), "spin")
        check(L.btc_spin(300 // CHAIN, b.cuda_stream), "spin")
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


for s in streams:       # first use in creation order
    check(L.btc_spin(1, s.cuda_stream), "spin")
torch.cuda.synchronize()
for rep in range(2 if CHAIN == 1 else 1):
    print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), "pass", rep)
    print("      " + " ".join("%4s" % n for n in names))
    for i, a in enumerate(streams):
        row = []
        for j, b in enumerate(streams):
            row.append("   ." if i == j else ("%4d" % int(pair(a, b) > 480) if CHAIN == 1 else "%4d" % int(pair(a, b))))
        print("%5s " % names[i] + " ".join(row))
