"""Python call sites of the small torch ops of one training step (fill_, zero_, copy_, add, mul, cat, ...): where the launches between our
kernels come from.   python tools/op_sites.py"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
from btcdet_amd.train_step import GroupOptimizer
from btcdet_amd.spconv import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
HEADS = os.environ.get("HEADS") or None   # HEADS=rpn|full: the heads behind the hot path too
cfg = load_cfg()
if os.environ.get("FEATURES") == "bf16":      # bench.py --features bf16
    cfg.MODEL.OCC.BACKBONE_3D["FEATURE_DTYPE"] = "bf16"
    cfg.MODEL.BACKBONE_3D["FEATURE_DTYPE"] = "bf16"
model = BtcHotPath(cfg, device=dev, heads=HEADS).to(dev).train()
occ = [p for p in model.occ_modules.parameters() if p.requires_grad]
det = [p for p in model.det_modules.parameters() if p.requires_grad]
opt = GroupOptimizer([dict(params=occ, lr=3e-3, weight_decay=1e-3, grad_norm_clip=10.0), dict(params=det, lr=1e-2, weight_decay=1e-2, grad_norm_clip=10.0)], 1000)
batches = bench.build_batches(2, 0, dev)
ops.set_defer_wgrad_join(True)
step = bench.make_step(model, model, model.dataset.data_processor, [opt], None, None, threaded=False, det_loss=model.det_loss)   # one thread: stacks are attributable
for i in range(4):
    step(batches[i % 2], batches[(i + 1) % 2])
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
WATCH = ("fill_", "zero_", "zeros", "zeros_like", "new_zeros", "full", "copy_", "_to_copy", "clone", "contiguous", "add", "add_", "mul", "mul_", "cat", "sum",
         "index", "neg", "div", "sub", "mean", "sqrt", "clamp", "where", "eq", "gt", "lt", "ne", "constant_pad_nd", "ones_like", "ones", "masked_fill_",
         "index_put_", "scatter_", "gather", "cumsum", "sort", "nonzero", "arange", "stack", "max", "min", "abs", "exp", "log", "sigmoid", "pow")
sites = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH or os.environ.get("ALL_OPS") == "1":
            fr = [f for f in traceback.extract_stack() if "/repo/" in f.filename and "op_sites" not in f.filename]
            site = "%s:%d %s" % (os.path.relpath(fr[-1].filename, ROOT), fr[-1].lineno, fr[-1].line) if fr else "(no repo frame: autograd thread / C++)"
            sites[(name, site[:130])] += 1
        return func(*args, **(kwargs or {}))


N = 2
with Census():
    for i in range(N):
        step(batches[i % 2], batches[(i + 1) % 2])
torch.cuda.synchronize()
for (name, site), c in sorted(sites.items(), key=lambda x: -x[1])[:int(os.environ.get("TOP", "70"))]:
    print("%5.1f /step  %-16s %s" % (c / N, name, site))

by_file = collections.Counter()
for (name, site), c in sites.items():
    by_file[site.split(":")[0]] += c
print("---- launches per step by file")
for f, c in by_file.most_common(15):
    print("%6.1f /step  %s" % (c / N, f))
