import os, torch, torch.distributed as dist, threading
print("before", len(os.sched_getaffinity(0)), torch.get_num_threads())
torch.cuda.set_device(0)
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29540")
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
dist.barrier()
print("after", len(os.sched_getaffinity(0)), torch.get_num_threads(), threading.active_count())
for k in sorted(os.environ):
    if any(t in k for t in ("NCCL","RCCL","HSA","HIP","OMP","ROC")): print(k, os.environ[k])
