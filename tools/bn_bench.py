"""BatchNorm(+ReLU) forward / backward launch pairs at the hot path's layer sizes, per workgroup-size setting of the statistics
passes (tuning keys BTC_TUNE_BN_FWD_KB / BTC_TUNE_BN_BWD_KB): us per call (stats + apply), HIP events, one stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from btcdet_amd._lib import lib, ptr, check, stream_ptr
dev = torch.device("cuda:0")
L = lib()
LAYERS = [(210000, 32), (210000, 5), (26000, 32), (12000, 32), (3000, 64), (40000, 16), (36000, 16), (42000, 32), (32000, 64), (17000, 64),
          (17000, 128), (12000, 128), (8000, 64)]
BF = os.environ.get("BF") == "1"
dt = torch.bfloat16 if BF else torch.float32


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


FWD = [0, 16, 32, 64, 128, 256]
BWD = [0, 32, 64, 128, 256, 512]
print("%8s %4s | fwd us at KB/workgroup " % ("N", "C") + " ".join("%6s" % (k or "dflt") for k in FWD) + " | bwd " + " ".join("%6s" % (k or "dflt") for k in BWD))
tot_f, tot_b = [0.0] * len(FWD), [0.0] * len(BWD)
for N, C in LAYERS:
    if BF and C % 4:
        continue
    x = torch.randn((N, C), device=dev).to(dt)
    dy = torch.randn((N, C), device=dev).to(dt)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    gamma, beta = torch.rand((C,), device=dev) + 0.5, torch.randn((C,), device=dev)
    rm, rv = torch.zeros((C,), device=dev), torch.ones((C,), device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    stats, dparam = torch.empty((2, C), device=dev), torch.empty((2, C), device=dev)
    need = L.btc_bn_ws_bytes(C)
    ws = torch.zeros((need,), dtype=torch.uint8, device=dev)
    fwd = L.btc_bn_relu_fwd_bf16 if BF else L.btc_bn_relu_fwd
    bwd = L.btc_bn_relu_bwd_bf16 if BF else L.btc_bn_relu_bwd
    f = lambda: check(fwd(ptr(x), N, C, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nbt), 0.01, 1e-3, 1, 1, ptr(y), ptr(stats[0]), ptr(stats[1]),
                          ptr(ws), need, stream_ptr()), "fwd")
    b = lambda: check(bwd(ptr(x), ptr(y), ptr(dy), N, C, ptr(gamma), ptr(stats[0]), ptr(stats[1]), 1, 1, ptr(dx), ptr(dparam[0]), ptr(dparam[1]),
                          ptr(ws), need, stream_ptr()), "bwd")
    tf, tb = [], []
    for k in FWD:
        check(L.btc_tune_set(9, k), "tune")
        tf.append(timed(f))
    check(L.btc_tune_set(9, 0), "tune")
    for k in BWD:
        check(L.btc_tune_set(10, k), "tune")
        tb.append(timed(b))
    check(L.btc_tune_set(10, 0), "tune")
    tot_f = [a + t for a, t in zip(tot_f, tf)]
    tot_b = [a + t for a, t in zip(tot_b, tb)]
    print("%8d %4d | %22s " % (N, C, "") + " ".join("%6.1f" % t for t in tf) + " |     " + " ".join("%6.1f" % t for t in tb))
print("%13s | %22s " % ("sum", "") + " ".join("%6.1f" % t for t in tot_f) + " |     " + " ".join("%6.1f" % t for t in tot_b))
