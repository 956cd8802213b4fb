"""cProfile of BtcHotPath.prepare alone (tiny scenes: host cost), cumulative and internal times"""
import cProfile, io, os, pstats, sys, time
os.environ.setdefault("BTC_BENCH_AZ_STEP", "4.0")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
dev = torch.device("cuda:0")
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
batches = bench.build_batches(2, 0, dev)
for i in range(5):
    model.prepare(batches[i % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50):
    model.prepare(batches[i % 2])
torch.cuda.synchronize()
print("prepare: %.3f ms per call (tiny scenes, incl. its read-backs)" % ((time.perf_counter() - t0) / 50 * 1e3))
pr = cProfile.Profile(); pr.enable()
for i in range(50):
    model.prepare(batches[i % 2])
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(sys.argv[1] if len(sys.argv) > 1 else "tottime").print_stats(28)
print(s.getvalue()[:6000])
