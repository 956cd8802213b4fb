"""Golden vectors of the reference's optimizer step (runs here only): its OWN build_optimizer('adam_onecycle') /
build_scheduler (tools/train_utils/optimization) and the loop body of train_one_epoch_multi_opt
(tools/train_utils/train_utils.py:121-124: clip_grad_norm_, optimizer.step(), lr_scheduler.step(it), LR_CLIP) on a small
module with seeded gradients -> tests/golden/optim.npz.

    python tests/golden/gen_optim_golden.py
"""
import collections
import collections.abc
import os
import sys

import numpy as np
import torch
from torch import nn

collections.Iterable = collections.abc.Iterable          # fastai_optim.py:3 (removed from `collections` in Python 3.10)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/tools")
import common  # noqa: E402
from train_utils.optimization import build_optimizer, build_scheduler  # noqa: E402
from torch.nn.utils import clip_grad_norm_  # noqa: E402


class Cfg(dict):
    __getattr__ = dict.__getitem__


def tiny():
    m = nn.Sequential(nn.Linear(6, 8), nn.BatchNorm1d(8), nn.ReLU(), nn.Linear(8, 4, bias=False), nn.BatchNorm1d(4))
    common.init_by_name(m)
    return m


def seeded_grads(m, it, big):
    for j, (n, p) in enumerate(m.named_parameters()):
        u = torch.from_numpy(common._hash01(p.numel(), 100 * it + j)).reshape(p.shape)
        p.grad = (u - 0.5) * (40.0 if big else 0.5)


if __name__ == "__main__":
    gold = {}
    total_it_each_epoch, epochs = 10, 4
    for tag, cfg in (("det", Cfg(OPTIMIZER="adam_onecycle", LR=0.01, WEIGHT_DECAY=0.01, MOMS=[0.95, 0.85], PCT_START=0.4, DIV_FACTOR=10,
                                 LR_CLIP=1e-7, GRAD_NORM_CLIP=10, LR_WARMUP=False, DECAY_STEP_LIST=[35, 45], LR_DECAY=0.1)),
                     ("occ", Cfg(OPTIMIZER="adam_onecycle", LR=0.003, WEIGHT_DECAY=0.001, MOMS=[0.95, 0.85], PCT_START=0.4, DIV_FACTOR=10,
                                 LR_CLIP=1e-7, GRAD_NORM_CLIP=10, LR_WARMUP=False, DECAY_STEP_LIST=[35, 45], LR_DECAY=0.1))):
        m = tiny()
        opt = build_optimizer(m, cfg)
        sched, _ = build_scheduler(opt, total_iters_each_epoch=total_it_each_epoch, total_epochs=epochs, last_epoch=-1, optim_cfg=cfg)
        lrs, moms, snaps = [], [], []
        for it in range(total_it_each_epoch * epochs + 3):       # runs past the end of the schedule too
            lrs.append(float(opt.lr))
            moms.append(float(opt.mom))
            opt.zero_grad()
            seeded_grads(m, it, big=(it % 3 == 0))               # every third step the norm clip bites
            clip_grad_norm_(m.parameters(), cfg.GRAD_NORM_CLIP)
            opt.step()
            sched.step(it)
            opt.lr = max(opt.lr, cfg.LR_CLIP)
            if it in (0, 1, 2, 15, 16, 17, 39, 42):
                snaps.append(np.concatenate([p.detach().numpy().reshape(-1) for p in m.parameters()]))
            if it == 16 and tag == "det":     # the reference's optimizer state in mid-run + its model state, for the resume test
                import copy
                torch.save({"epoch": 1, "it": 17, "model_state": copy.deepcopy(m.state_dict()), "optimizer_state_lst": [copy.deepcopy(opt.state_dict())],
                            "version": "reference"}, os.path.join(HERE, "optim_resume_ref.pth"))
        gold[tag + "_lr"], gold[tag + "_mom"], gold[tag + "_params"] = np.array(lrs), np.array(moms), np.stack(snaps)
    gold["snap_iters"] = np.array([0, 1, 2, 15, 16, 17, 39, 42])
    gold["meta"] = np.array([total_it_each_epoch, epochs])
    np.savez_compressed(os.path.join(HERE, "optim.npz"), **gold)
    print({k: v.shape for k, v in gold.items()})
