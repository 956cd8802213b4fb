"""Weight-gradient launches of one bench step, one by one (HIP events, one stream): us, algorithmic GB/s and TFLOP/s, and the
same launch without its MFMA phase / without its gathers (BTC_TUNE_APPLY_DEBUG 4 / 8: wrong results, timing only) -- which
side of the kernel bounds it."""
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


def timed(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("%7s %7s %3s %4s %4s %8s | %7s %6s %6s | two-barrier kernel: %7s %8s %8s  same bits" % ("n_res", "n_src", "K", "cin", "cout", "pairs", "us", "GB/s", "TF/s", "us", "no-mfma", "no-gath"))
tot = np.zeros(4)
for (f, w, b, mf, mb) in cap:
    cin, cout, K = w.shape[-2], w.shape[-1], mf.shape[1]
    n_res, n_src = mf.shape[0], mb.shape[0]
    pairs = int((mf >= 0).sum())
    g = torch.randn((n_res, cout), device=dev).to(f.dtype)
    bf = f.dtype == torch.bfloat16
    wg = L.btc_conv_wgrad_bf16 if bf else L.btc_conv_wgrad
    wsb = L.btc_conv_wgrad_ws_bytes(n_res, K, cin, cout, n_src)
    check(L.btc_tune_set(11, 1), "t")
    wsb = max(wsb, L.btc_conv_wgrad_ws_bytes(n_res, K, cin, cout, n_src))   # the two kernels split the rows differently
    check(L.btc_tune_set(11, 0), "t")
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    dw = torch.empty_like(w)
    fn = lambda: check(wg(ptr(f), ptr(g), ptr(mf), n_res, ptr(mb), n_src, K, cin, cout, ptr(dw), ptr(ws), wsb, stream_ptr()), "w")
    ts = [timed(fn)]
    ref = dw.clone()
    check(L.btc_tune_set(11, 1), "t")   # the two-barrier kernel, and its timing-only variants
    for dbg in (0, 4, 8):
        check(L.btc_tune_set(3, dbg), "t")
        ts.append(timed(fn))
        if dbg == 0:
            same = bool(torch.equal(ref, dw))
    check(L.btc_tune_set(3, 0), "t")
    check(L.btc_tune_set(11, 0), "t")
    tot += np.array(ts)
    nbytes, flops = (2 if bf else 4) * pairs * (cin + cout) + 4 * K * cin * cout, 2 * pairs * cin * cout
    print("%7d %7d %3d %4d %4d %8d | %7.1f %6.0f %6.2f | %26.1f %8.1f %8.1f  %s" % (n_res, n_src, K, cin, cout, pairs, ts[0], nbytes / ts[0] / 1e3, flops / ts[0] / 1e6,
                                                                                 ts[1], ts[2], ts[3], same))
print("totals us: %.0f; two-barrier kernel %.0f, no MFMA %.0f, no gathers %.0f" % tuple(tot))
