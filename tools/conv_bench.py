"""A/B timing of the sparse-conv apply kernel variants on the real layers of one hot-path step (tuning helper).

usage: python tools/conv_bench.py [variant ...]   variant = kernel:nt:xcd (btc_tune_set values; 0 = built-in policy)
Every variant's output is compared bit-for-bit with the first one."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from btcdet_amd import _lib
from btcdet_amd._lib import ptr, stream_ptr, check
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
from btcdet_amd.spconv import ops

def parse_variant(v):
    """kernel:nt:xcd positional btc_tune_set values, or key=value items (13=5: ring depth 5), mixed"""
    vals = [0] * 24
    if v.startswith("split"):     # split[:tune...]: the split-operand kernel where it applies (operands = 3), tune values after the colon
        vals.append(1)
        v = v[6:] or "0"
    for i, part in enumerate(v.split(":")):
        if "=" in part:
            k, x = part.split("=")
            vals[int(k)] = int(x)
        else:
            vals[i] = int(part)
    return tuple(vals)


variants = [parse_variant(v) for v in sys.argv[1:]] or [parse_variant("0"), parse_variant("2")]
names = sys.argv[1:] or ["0", "2"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
opts = [torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3)]
batches = bench.build_batches(2, 0, dev)
step = bench.make_step(model, model, model.dataset.data_processor, opts)
for _ in range(int(os.environ.get("CB_WARM", "1"))):   # the occupancy head's predictions (hence the detection levels' sizes) settle over the first ~10 steps
    step(batches[0])
ops.CAPTURE = []
step(batches[0])
cap, ops.CAPTURE = ops.CAPTURE, None
torch.cuda.synchronize()
L = _lib.lib()
_scratch = torch.empty(48 << 20, dtype=torch.uint8, device=dev)      # z-split launches of the split-operand kernel
check(L.btc_set_scratch(stream_ptr(), ptr(_scratch), _scratch.numel()), "btc_set_scratch")


def tune(v):
    v = tuple(v[:24]) + (0,) * (24 - len(v[:24]))
    for key, val in enumerate(v):
        check(L.btc_tune_set(key, val), "btc_tune_set")


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


seen = set()
print("%-5s %7s %7s %2s %4s %4s %8s | " % ("dir", "n_res", "n_src", "K", "cred", "cres", "pairs") +
      " ".join("%12s" % n for n in names))
tot = np.zeros(len(variants))
ONLY = os.environ.get("CB_ONLY")  # e.g. "256,128" = only layers with these (cin, cout)
for feats, w, b, mf, mb in cap:
    K = mf.shape[1]
    if ONLY and (w.shape[-2], w.shape[-1]) not in [tuple(int(x) for x in o.split(",")) for o in ONLY.split(";")]:
        continue
    cin, cout = w.shape[-2], w.shape[-1]
    pairs = int((mf >= 0).sum())
    key = (mf.shape[0], mb.shape[0], K, cin, cout, pairs)
    mult = sum(1 for c in cap if (c[3].shape[0], c[4].shape[0], c[3].shape[1], c[1].shape[-2], c[1].shape[-1]) == key[:5])
    if key in seen:
        continue
    seen.add(key)
    n_res, n_src = mf.shape[0], mb.shape[0]
    out = torch.empty((n_res, cout), device=dev)
    dout = torch.randn((n_res, cout), device=dev)
    din = torch.empty((n_src, cin), device=dev)
    dw = torch.empty_like(w)
    ws_cache = {}
    for direction in os.environ.get("CB_DIRS", "fwd,dgrad").split(","):
        us, ref = [], None
        for v in variants:
            tune(v)
            split = len(v) > 24 and direction != "wgrad" and L.btc_conv_split_supported(K, cin if direction == "fwd" else cout, cout if direction == "fwd" else cin) == 1
            if split:
                if "planes" not in ws_cache:
                    q = torch.empty((2, 3) + tuple(w.shape), dtype=torch.bfloat16, device=dev)
                    check(L.btc_weights_split3(ptr(w), K, cin, cout, ptr(q[0]), ptr(q[1]), stream_ptr()), "split3")
                    ws_cache["planes"] = q
                q = ws_cache["planes"]
                if direction == "fwd":
                    fn = lambda: check(L.btc_conv_apply_src(0, 3, ptr(feats), int(feats.shape[0]), ptr(q[1]), ptr(b), ptr(mf), None, n_res, K, cin, cout, ptr(out), stream_ptr()), "fwd split")
                    res = out
                else:
                    fn = lambda: check(L.btc_conv_apply_src(1, 3, ptr(dout), int(dout.shape[0]), ptr(q[0]), None, ptr(mb), None, n_src, K, cin, cout, ptr(din), stream_ptr()), "dgrad split")
                    res = din
            elif direction == "fwd":
                fn = lambda: check(L.btc_conv_fwd(ptr(feats), ptr(w), ptr(b), ptr(mf), n_res, K, cin, cout, ptr(out), stream_ptr()), "fwd")
                res = out
            elif direction == "wgrad":
                nb = L.btc_conv_wgrad_ws_bytes(n_res, K, cin, cout, n_src)
                ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
                fn = lambda: check(L.btc_conv_wgrad(ptr(feats), ptr(dout), ptr(mf), n_res, ptr(mb), n_src, K, cin, cout, ptr(dw), ptr(ws), nb,
                                                    stream_ptr()), "wgrad")
                res = dw
            else:
                fn = lambda: check(L.btc_conv_dgrad(ptr(dout), ptr(w), ptr(mb), n_src, K, cin, cout, ptr(din), stream_ptr()), "dgrad")
                res = din
            res.zero_()
            t = timed(fn)
            cur = res.clone()
            if ref is None:
                ref = cur
            if os.environ.get("CB_F64") and direction == "fwd" and n_res <= 50000 and (split or v is variants[0]):
                f64, w64 = feats.double(), w.double().reshape(K, cin, cout)
                r64 = torch.zeros((n_res, cout), dtype=torch.float64, device=dev)
                for kk in range(K):
                    idx = mf[:, kk].long()
                    r64 += (f64[idx.clamp(min=0)] * (idx >= 0).unsqueeze(1)) @ w64[kk]
                if b is not None:
                    r64 += b.double()
                print("      vs fp64: %-12s max |err| / scale %.2e   rms err / rms %.2e" % (
                    "split" if split else "exact chain", float((cur.double() - r64).abs().max() / r64.abs().max()),
                    float((cur.double() - r64).pow(2).mean().sqrt() / r64.pow(2).mean().sqrt())))
            if split:
                err = float((ref - cur).abs().max() / ref.abs().max())
                ok = err <= 2e-6
                worst_split = max(globals().get("worst_split", 0.0), err)
            elif direction == "wgrad":
                ok = bool((ref - cur).abs().max() <= 1e-4 * ref.abs().max())
            else:
                ok = torch.equal(ref.view(torch.int32), cur.view(torch.int32)) or (len(v) > 3 and v[3])
            us.append((t, ok))
        tot += np.array([u[0] for u in us]) * mult
        cred, cres = (cin, cout) if direction == "fwd" else (cout, cin)
        print("%-5s %7d %7d %2d %4d %4d %8d | " % (direction, n_res, n_src, K, cred, cres, pairs) +
              " ".join("%9.1f%s" % (t, "   " if ok else " !!") for t, ok in us) + "  x%d" % mult)
tune(())
print("worst split-operand error relative to the scale: %.2e" % globals().get("worst_split", 0.0))
print("sum over the step's layers (us): " + " ".join("%12.0f" % t for t in tot))
