"""Where the three host threads of the pipelined step spend their time: a sampling profiler in the process (sys._current_frames every
~0.4 ms from a fourth thread), per thread the functions by inclusive and by leaf samples.  The sampler needs the interpreter lock like
everybody else, so it sees the threads at the moments one of them lets go of it -- biased towards the calls that release it (kernel
launches, copies), still the cheapest view of "which Python is in the step" there is on a box without py-spy.
usage: python tools/host_sampler.py [steps=400] [top=28]"""
import collections, os, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from btcdet_amd.affinity import pin_to_gpu
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
from btcdet_amd.trainer import HotPathTrainer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
top = int(sys.argv[2]) if len(sys.argv) > 2 else 28
dev = torch.device("cuda", 0); torch.cuda.set_device(0); pin_to_gpu(0, 0, 1)
torch.manual_seed(666); np.random.seed(666)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
tr = HotPathTrainer(model, det_loss=model.det_loss)
pool = bench.build_batches(96, 0, dev, 2, "kitti")
for i in range(32):
    tr.step(pool[i % 96], pool[(i + 1) % 96])
torch.cuda.synchronize()

incl = collections.defaultdict(collections.Counter)
leaf = collections.defaultdict(collections.Counter)
total = collections.Counter()
stop = [False]
me = [None]


def label(f):
    co = f.f_code
    return "%s:%d %s" % (os.path.relpath(co.co_filename, ROOT) if co.co_filename.startswith(ROOT) else os.path.basename(co.co_filename), co.co_firstlineno, co.co_name)


def sampler():
    me[0] = threading.get_ident()
    names = {}
    while not stop[0]:
        for t in threading.enumerate():
            names[t.ident] = t.name
        for tid, f in sys._current_frames().items():
            if tid == me[0]:
                continue
            name = names.get(tid, str(tid))
            total[name] += 1
            leaf[name][label(f) + " @%d" % f.f_lineno] += 1
            seen = set()
            while f is not None:
                l = label(f)
                if l not in seen:
                    incl[name][l] += 1
                    seen.add(l)
                f = f.f_back
        time.sleep(0.0004)


th = threading.Thread(target=sampler, name="sampler", daemon=True)
th.start()
t0 = time.perf_counter()
for i in range(32, 32 + n):
    tr.step(pool[i % 96], pool[(i + 1) % 96])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
stop[0] = True
th.join()
print("%d steps, %.2f ms per step (with the sampler running)" % (n, 1e3 * dt / n))
for name in sorted(total, key=lambda k: -total[k]):
    if total[name] < 20:
        continue
    print("\n==== thread %s: %d samples" % (name, total[name]))
    print("  -- inclusive")
    for l, c in incl[name].most_common(top):
        print("  %5.1f %%  %s" % (100.0 * c / total[name], l))
    print("  -- leaf (file:first line function @line)")
    for l, c in leaf[name].most_common(top):
        print("  %5.1f %%  %s" % (100.0 * c / total[name], l))
tr.finish()
