"""row-order launch with and without the first-offset keys (A/B probe)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from btcdet_amd import _lib
L = _lib.lib(); ptr = _lib.ptr
torch.manual_seed(0)
for n in (25865, 209797):
    K = 27
    nbr = torch.where(torch.rand(n, K, device="cuda") < 0.11, torch.randint(0, n, (n, K), device="cuda", dtype=torch.int32), torch.full((n, K), -1, device="cuda", dtype=torch.int32)).contiguous()
    has = nbr >= 0
    first = torch.where(has.any(1), has.int().argmax(1), torch.full((n,), K, device="cuda")).int().contiguous()
    order = torch.empty(n, dtype=torch.int32, device="cuda"); order2 = torch.empty_like(order)
    ns = (ctypes.c_int32 * 1)(n); ks = (ctypes.c_int32 * 1)(K); ps = (ctypes.c_void_p * 1)(ptr(nbr)); fs = (ctypes.c_void_p * 1)(ptr(first))
    for name, fn in (("map", lambda: L.btc_row_orders(ps, ns, ks, 1, ptr(order), _lib.stream_ptr())),
                     ("keyed", lambda: L.btc_row_orders_keyed(ps, fs, ns, ks, 1, ptr(order2), _lib.stream_ptr()))):
        for _ in range(3): assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(n, name, "%.1f us" % (1e3 * e0.elapsed_time(e1) / 20))
    print("equal", bool(torch.equal(order, order2)))
