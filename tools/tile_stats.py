"""MFMA row-slots per real (row, offset) pair of every distinct neighbour map of one hot-path step, for the map order, the
order hint the step uses, and rows sorted by their 27-bit presence mask inside 2048-row / 512-row blocks (what a mask-keyed
row-order hint would give the 16-row wave tiles of conv_apply_g)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from btcdet_amd.btc_path import BtcHotPath
from btcdet_amd.config import load_cfg
from btcdet_amd.spconv import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = BtcHotPath(load_cfg(), device=dev).to(dev).train()
opts = [torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3)]
batches = bench.build_batches(2, 0, dev)
step = bench.make_step(model, model, model.dataset.data_processor, opts)
step(batches[0])
ops.CAPTURE = []
step(batches[0])
cap, ops.CAPTURE = ops.CAPTURE, None
torch.cuda.synchronize()


def work(act, order=None):
    a = act if order is None else act[order]
    n, K = a.shape
    pad = (-n) % 16
    if pad:
        a = torch.cat([a, torch.zeros((pad, K), dtype=torch.bool, device=a.device)])
    return float(a.view(-1, 16, K).any(1).sum() * 16) / max(float(act.sum()), 1.0)


def block_sort(key, blk):
    n = key.shape[0]
    blk_id = torch.arange(n, device=key.device) // blk
    return torch.argsort(blk_id * (1 << 32) + key, stable=True)


seen = set()
for feats, w, b, mf, mb in cap:
    for name, m in (("fwd", mf), ("bwd", mb)):
        if m is None or m.numel() == 0 or (m.data_ptr(), m.shape) in seen:
            continue
        seen.add((m.data_ptr(), m.shape))
        n, K = m.shape
        act = m >= 0
        mask = (act.long() << torch.arange(K, device=dev)).sum(1)
        print("%-4s rows %7d K %2d pairs/row %5.2f | slots/pair: map order %.2f  mask/2048 %.2f  mask/512 %.2f  mask/128 %.2f  mask/all %.2f" % (
            name, n, K, float(act.sum()) / n, work(act), work(act, block_sort(mask, 2048)), work(act, block_sort(mask, 512)),
            work(act, block_sort(mask, 128)), work(act, torch.argsort(mask, stable=True))))
