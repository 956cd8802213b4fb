"""host / device cost of one all-reduce at world size 1: the reducer's own RCCL communicator vs the process group (A/B probe)"""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from btcdet_amd.rccl_direct import RcclComm
comm = RcclComm(dev)
cs = torch.cuda.Stream()
for n in (1 << 18, 2_600_000):
    x = torch.ones(n, device=dev)
    for name, fn in (("direct", lambda: comm.all_reduce_(x, cs, True)), ("direct_sum", lambda: comm.all_reduce_(x, cs, False)),
                     ("torch", lambda: dist.all_reduce(x, op=dist.ReduceOp.AVG, async_op=True))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(cs)
        for _ in range(50):
            fn()
        e1.record(cs)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%-10s n=%8d host %.1f us/call, cs-stream %.1f us/call, wall incl. sync %.1f us/call, x[0]=%g" % (name, n, 1e6 * (t1 - t0) / 50, 1e3 * e0.elapsed_time(e1) / 50, 1e6 * (t2 - t0) / 50, float(x[0])))
dist.destroy_process_group()
