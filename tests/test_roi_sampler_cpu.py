"""The ROI-head proposal sampler (btcdet_amd/roi_targets.sample_rois: the reference's subsample_rois / sample_bg_inds,
proposal_target_layer.py:117-197, as fixed-shape tensor arithmetic) on CPU tensors: quotas, candidate sets and the degenerate cases,
over many random overlap vectors.  (The IoU matching and everything else of the layer is GPU work: tests/test_hip_roi_targets.py.)"""
import numpy as np
import pytest
import torch


class _Cfg(dict):
    __getattr__ = dict.__getitem__


CFG = _Cfg(ROI_PER_IMAGE=128, FG_RATIO=0.5, REG_FG_THRESH=0.55, CLS_FG_THRESH=0.75, CLS_BG_THRESH=0.25, CLS_BG_THRESH_LO=0.1, HARD_BG_RATIO=0.8)


def _check(ov, sel):
    R, quota = CFG.ROI_PER_IMAGE, 64
    fg = ov >= 0.55
    easy = ov < 0.1
    hard = ~fg & ~easy
    n_fg, n_easy, n_hard = int(fg.sum()), int(easy.sum()), int(hard.sum())
    assert sel.shape == (R,) and sel.min() >= 0 and sel.max() < ov.shape[0]
    if n_fg and (n_easy + n_hard):
        k = min(quota, n_fg)
        assert fg[sel[:k]].all() and len(set(sel[:k].tolist())) == k and not fg[sel[k:]].any()
        m = R - k
        h = min(int(m * 0.8), n_hard) if (n_hard and n_easy) else (m if n_hard else 0)
        assert hard[sel[k:k + h]].all() and easy[sel[k + h:]].all()
    elif n_fg:
        assert fg[sel].all()
    else:
        h = min(int(R * 0.8), n_hard) if (n_hard and n_easy) else (R if n_hard else 0)
        assert hard[sel[:h]].all() and easy[sel[h:]].all()


@pytest.mark.parametrize("seed", range(12))
def test_sampler_quotas(seed):
    from btcdet_amd.roi_targets import sample_rois
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 600))
    kind = seed % 6
    ov = rng.uniform(0, 1, n)
    if kind == 1:
        ov = rng.uniform(0.6, 1, n)            # foreground only
    elif kind == 2:
        ov = rng.uniform(0, 0.09, n)           # easy background only
    elif kind == 3:
        ov = rng.uniform(0.1, 0.5, n)          # hard background only
    elif kind == 4:
        ov = np.where(rng.uniform(0, 1, n) < 0.02, 0.9, rng.uniform(0, 0.5, n))    # very few foreground
    elif kind == 5:
        ov = np.where(rng.uniform(0, 1, n) < 0.9, 0.9, 0.3)                        # very few hard background, no easy
    ov_t = torch.from_numpy(ov.astype(np.float32))
    gen = torch.Generator().manual_seed(seed)
    a = sample_rois(ov_t, CFG, gen).numpy()
    b = sample_rois(ov_t, CFG, gen).numpy()
    _check(ov.astype(np.float32), a)
    _check(ov.astype(np.float32), b)
    if n > 200 and kind == 0:
        assert not np.array_equal(a, b)
