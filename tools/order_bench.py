"""gain of the row-order hint per captured conv launch of one bench step (fwd / dgrad / wgrad), bit-equality of fwd / dgrad"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
from btcdet_amd.spconv import ops
from btcdet_amd.spconv.ops import lib, ptr, check, stream_ptr
from btcdet_amd.train_step import GroupOptimizer
dev = torch.device("cuda:0")
torch.manual_seed(0)
BF = os.environ.get("BF") == "1"
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
if BF:
    model.set_feature_dtype(torch.bfloat16) if hasattr(model, "set_feature_dtype") else None
opt = GroupOptimizer([dict(params=[p for p in model.parameters() if p.requires_grad], lr=1e-3)], 1000)
batches = bench.build_batches(2, 0, dev)
step = bench.make_step(model, model, model.dataset.data_processor, [opt])
for i in range(2):
    step(batches[i % 2])
torch.cuda.synchronize()
ops.PROFILE = ops.LaunchProfile()   # python route so CAPTURE sees every conv
ops.CAPTURE = []
step(batches[0])
torch.cuda.synchronize()
cap, ops.CAPTURE, ops.PROFILE = ops.CAPTURE, None, None
L = lib()

def orders(maps):
    return ops.row_orders(maps)

def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

# one sort for all distinct maps of the step
uniq = {}
for (f, w, b, mf, mb) in cap:
    for m in (mf, mb):
        if m.shape[1] <= 64:
            uniq.setdefault(m.data_ptr(), m)
maps = list(uniq.values())
t_sort = timeit(lambda: orders(maps), 5)
ords = dict(zip([m.data_ptr() for m in maps], orders(maps)))
for m in maps:   # a permutation?
    o = ords[m.data_ptr()]
    assert torch.equal(torch.sort(o.long())[0], torch.arange(m.shape[0], device=dev)), "not a permutation"
print("sort of %d maps, %d rows: %.1f us" % (len(maps), sum(m.shape[0] for m in maps), t_sort))
tot = np.zeros(6)
print("%7s %7s %3s %4s %4s | %7s %7s | %7s %7s | %7s %7s" % ("n_res", "n_src", "K", "cin", "cout", "fwd", "fwd_o", "dgrad", "dgrad_o", "wgrad", "wgrad_o"))
for (f, w, b, mf, mb) in cap:
    cin, cout = w.shape[-2], w.shape[-1]
    K = mf.shape[1]
    n_res, n_src = mf.shape[0], mb.shape[0]
    of, ob = ords.get(mf.data_ptr()), ords.get(mb.data_ptr())
    if of is None or ob is None:
        continue
    bf = f.dtype == torch.bfloat16
    opnd = 1 if bf else 0
    out0 = torch.empty((n_res, cout), dtype=f.dtype, device=dev)
    out1 = torch.empty_like(out0)
    fw = lambda o, dst: check(L.btc_conv_apply_ordered(0, opnd, ptr(f), ptr(w), ptr(b), ptr(mf), ptr(o), n_res, K, cin, cout, ptr(dst), stream_ptr()), "f")
    t0 = timeit(lambda: fw(None, out0)); t1 = timeit(lambda: fw(of, out1))
    assert torch.equal(out0, out1), "fwd differs"
    g = torch.randn((n_res, cout), device=dev).to(f.dtype)
    d0 = torch.empty((n_src, cin), dtype=f.dtype, device=dev); d1 = torch.empty_like(d0)
    dg = lambda o, dst: check(L.btc_conv_apply_ordered(1, opnd, ptr(g), ptr(w), None, ptr(mb), ptr(o), n_src, K, cin, cout, ptr(dst), stream_ptr()), "d")
    t2 = timeit(lambda: dg(None, d0)); t3 = timeit(lambda: dg(ob, d1))
    assert torch.equal(d0, d1), "dgrad differs"
    wsb = L.btc_conv_wgrad_ws_bytes(n_res, K, cin, cout, n_src)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    w0 = torch.empty_like(w); w1 = torch.empty_like(w)
    wg = lambda o1, o2, dst: check(L.btc_conv_wgrad_ordered(opnd, ptr(f), ptr(g), ptr(mf), n_res, ptr(mb), n_src, ptr(o1), ptr(o2), K, cin, cout, ptr(dst), ptr(ws), wsb, stream_ptr()), "w")
    t4 = timeit(lambda: wg(None, None, w0)); t5 = timeit(lambda: wg(of, ob, w1))
    err = ((w0 - w1).abs().max() / w0.abs().max().clamp_min(1e-20)).item()
    tot += np.array([t0, t1, t2, t3, t4, t5])
    print("%7d %7d %3d %4d %4d | %7.1f %7.1f | %7.1f %7.1f | %7.1f %7.1f  wgrad rel %.1e" % (n_res, n_src, K, cin, cout, t0, t1, t2, t3, t4, t5, err))
print("totals us: fwd %.0f -> %.0f, dgrad %.0f -> %.0f, wgrad %.0f -> %.0f" % tuple(tot))
