"""the occupancy net's transposed layers (deconv4 / deconv5) on one bench batch: dgrad launch time in map order and with the row-order
hint, exact and split kernels (what a row-order hint does to a DENSE backward map)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from btcdet_amd import _lib
from btcdet_amd._lib import check, ptr, stream_ptr
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
from btcdet_amd.spconv import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
bd = model.prepare(bench.build_batches(1, 0, dev)[0])
L = _lib.lib()
rbs = [rb for rb in bd.get("__occ_rulebooks__", [])] if "__occ_rulebooks__" in bd else []
cache = None
for k, v in bd.items():
    if isinstance(v, dict) and "__geometry_cache__" in v:
        cache = v
found = []
def walk(o, depth=0):
    if isinstance(o, ops.Rulebook):
        found.append(o)
    elif isinstance(o, dict) and depth < 4:
        for v in o.values():
            walk(v, depth + 1)
    elif isinstance(o, (list, tuple)) and depth < 4:
        for v in o:
            walk(v, depth + 1)
walk(bd)
seen = set()
def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for rb in found:
    if id(rb) in seen or rb.mode != ops.MODE_TRANSPOSE:
        continue
    seen.add(id(rb))
    n_out, n_in, K = rb.nbr_out.shape[0], rb.map_bwd.shape[0], rb.K
    c = 32
    w = torch.randn((K, c, c), device=dev) * 0.05
    q = torch.empty((2, 3 * w.numel()), dtype=torch.bfloat16, device=dev)
    check(L.btc_weights_split3(ptr(w), K, c, c, ptr(q[0]), ptr(q[1]), stream_ptr()), "s")
    dout = torch.randn((n_out, c), device=dev)
    din = torch.empty((n_in, c), device=dev)
    order = ops.row_orders([rb.nbr_out, rb.map_bwd])[1]
    print("transposed rulebook: n_out %d n_in %d pairs/in-row %.1f  rb.order_in is None: %s" % (n_out, n_in, float((rb.map_bwd >= 0).sum()) / n_in, rb.order_in is None))
    for name, op, W in (("exact", 0, w), ("split", 3, q[0])):
        for oname, o in (("map order", None), ("hint", order)):
            t = timed(lambda: check(L.btc_conv_apply_ordered(1, op, ptr(dout), ptr(W), None, ptr(rb.map_bwd), ptr(o), n_in, K, c, c, ptr(din), stream_ptr()), "d"))
            print("   dgrad %-6s %-10s %7.1f us" % (name, oname, t))
