"""Weight-gradient launches of one bench step, one by one (HIP events, one stream): us, algorithmic GB/s and TFLOP/s, and the
same launch on the fp32-pipe kernels (BTC_TUNE_WGRAD_X = 1: conv_wgrad_rows_p / conv_wgrad_partial_p)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
from btcdet_amd.spconv import ops
from btcdet_amd._lib import lib, ptr, check, stream_ptr
from btcdet_amd.train_step import GroupOptimizer
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = load_cfg()
BF = os.environ.get("BF") == "1"   # bf16 activations (bench.py --features bf16)
if BF:
    cfg.MODEL.OCC.BACKBONE_3D["FEATURE_DTYPE"] = "bf16"
    cfg.MODEL.BACKBONE_3D["FEATURE_DTYPE"] = "bf16"
model = BtcHotPath(cfg, device=dev).to(dev).train()
opt = GroupOptimizer([dict(params=[p for p in model.parameters() if p.requires_grad], lr=1e-3)], 1000)
batches = bench.build_batches(2, 0, dev)
step = bench.make_step(model, model, model.dataset.data_processor, [opt])
step(batches[0])
ops.PROFILE = ops.LaunchProfile(); ops.CAPTURE = []
step(batches[0]); torch.cuda.synchronize()
cap, ops.CAPTURE, ops.PROFILE = ops.CAPTURE, None, None
L = lib()
ALT = tuple(int(v) for v in os.environ.get("ALT", "18=1").split("="))   # the tuning key = value of the second column (ALT=20=1: one item in flight)


def timed(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("%7s %7s %3s %4s %4s %8s | %9s %6s %6s | second column (tune %d = %d; default: the fp32-pipe kernels): %7s  rel. diff" % ("n_res", "n_src", "K", "cin", "cout", "pairs", "us", "GB/s", "TF/s", ALT[0], ALT[1], "us"))
tot = np.zeros(2)
for (f, w, b, mf, mb) in cap:
    cin, cout, K = w.shape[-2], w.shape[-1], mf.shape[1]
    if os.environ.get("NARROW") == "1" and cout > 8: continue   # (only the layers conv_wgrad_n takes)
    n_res, n_src = mf.shape[0], mb.shape[0]
    pairs = int((mf >= 0).sum())
    g = torch.randn((n_res, cout), device=dev).to(f.dtype)
    bf = f.dtype == torch.bfloat16
    wg = L.btc_conv_wgrad_bf16 if bf else L.btc_conv_wgrad
    wsb = L.btc_conv_wgrad_ws_bytes(n_res, K, cin, cout, n_src)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    dw = torch.empty_like(w)
    pmb = ptr(mb)    # (a submanifold rulebook has one map: the same pointer twice)
    fn = lambda: check(wg(ptr(f), ptr(g), ptr(mf), n_res, pmb, n_src, K, cin, cout, ptr(dw), ptr(ws), wsb, stream_ptr()), "w")
    ts = [timed(fn)]
    ref = dw.clone()
    check(L.btc_tune_set(ALT[0], ALT[1]), "t")   # default: the fp32-pipe kernels (conv_wgrad_rows_p / conv_wgrad_partial_p)
    ts.append(timed(fn))
    check(L.btc_tune_set(ALT[0], 0), "t")
    diff = float((ref - dw).abs().max() / (dw.abs().max() + 1e-30))
    tot += np.array(ts)
    nbytes, flops = (2 if bf else 4) * pairs * (cin + cout) + 4 * K * cin * cout, 2 * pairs * cin * cout
    print("%7d %7d %3d %4d %4d %8d | %9.1f %6.0f %6.2f | %46.1f  %.1e" % (n_res, n_src, K, cin, cout, pairs, ts[0], nbytes / ts[0] / 1e3, flops / ts[0] / 1e6, ts[1], diff))
print("totals us: %.0f; second column %.0f" % tuple(tot))
