"""conv_apply_g time vs row count for each wave shape / kc (SubM map of the bench scene's largest det level, truncated)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
from btcdet_amd.spconv import ops
from btcdet_amd._lib import lib, ptr, check, stream_ptr
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
batches = bench.build_batches(2, 0, dev)
from btcdet_amd.train_step import GroupOptimizer
opt = GroupOptimizer([dict(params=[p for p in model.parameters() if p.requires_grad], lr=1e-3)], 1000)
step = bench.make_step(model, model, model.dataset.data_processor, [opt])
step(batches[0])
ops.PROFILE = ops.LaunchProfile(); ops.CAPTURE = []
step(batches[0]); torch.cuda.synchronize()
cap, ops.CAPTURE, ops.PROFILE = ops.CAPTURE, None, None
L = lib()
big = max((c for c in cap if c[3].shape[0] == c[4].shape[0] and c[3].shape[0] < 100000 and c[3].shape[1] == 27), key=lambda c: c[3].shape[0])[3]
n0 = big.shape[0]
print("base map rows", n0, "pairs/row %.2f" % ((big >= 0).sum().item() / n0))

def timed(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

BF = os.environ.get("BF") == "1"
LAYERS = [(32, 32), (32, 64), (64, 64), (128, 128), (256, 128)]
SHAPES = {32: [221, 412, 411], 64: [422, 222, 141, 421, 241, 414], 128: [424, 242, 422, 224, 222, 418]}
for cin, cout in LAYERS:
    w = torch.randn((27, cin, cout), device=dev) * 0.05
    cands = [(s, kc) for s in SHAPES[cout] for kc in (32, 64) if cin % kc == 0]
    print("\n%d -> %d   n_rows | " % (cin, cout) + " ".join("%8s" % ("%d/%d" % c) for c in cands) + " | best")
    for n in list(range(4000, n0, 2000)) + [n0]:
        m = big[:n].clone(); m[m >= n] = -1
        f = torch.randn((n, cin), device=dev)
        out = torch.empty((n, cout), device=dev)
        ts = []
        for s, kc in cands:
            check(L.btc_tune_set(1, s), "t"); check(L.btc_tune_set(4, kc), "t")
            try:
                ts.append(timed(lambda: check(L.btc_conv_fwd(ptr(f), ptr(w), None, ptr(m), n, 27, cin, cout, ptr(out), stream_ptr()), "f")))
            except Exception as e:
                ts.append(float("nan"))
        check(L.btc_tune_set(1, 0), "t"); check(L.btc_tune_set(4, 0), "t")
        t_def = timed(lambda: check(L.btc_conv_fwd(ptr(f), ptr(w), None, ptr(m), n, 27, cin, cout, ptr(out), stream_ptr()), "f"))
        b = int(np.nanargmin(ts))
        print("%18d | " % n + " ".join("%8.1f" % t for t in ts) + " | %d/%d  default %.1f" % (cands[b][0], cands[b][1], t_def))
