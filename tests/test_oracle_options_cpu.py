"""The oracle's restatement of the options the configured model leaves off -- REVERSE_VIS = VCC / BACK_TRACK, OCC.DROPOUT_RATE > 0
with and without DROPOUT_RMV -- against vectors produced by the REAL reference's OccTargets3D (tests/golden/gen_options_golden.py ->
occ_options.npz): every mask bit for bit, both weight maps exactly, the dropped payload by SHA-1."""
import copy
import os

import numpy as np
import pytest
import torch

from golden_batch import golden_batch
from oracle import occ_oracle

from btcdet_amd.config import load_cfg

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "occ_options.npz"))
MASKS = ["occ_voxelwise_mask", "general_cls_loss_mask", "pos_mask", "general_reg_loss_mask", "occ_fore_cls_mask"]
CASES = {"vcc": dict(reverse_vis="VCC"), "back_track": dict(reverse_vis="BACK_TRACK"), "drop": dict(dropout=0.3), "drop_rmv": dict(dropout=0.3, rmv=True)}


def option_cfg(reverse_vis="NOTHING", dropout=0.0, rmv=False):
    cfg = copy.deepcopy(load_cfg())
    cfg.MODEL.OCC.PARAMS["REVERSE_VIS"] = reverse_vis
    cfg.DATA_CONFIG.OCC["DROPOUT_RATE"] = dropout
    cfg.DATA_CONFIG.OCC["DROPOUT_RMV"] = rmv
    return cfg


def dense(tag, key, shape):
    out = np.zeros(int(np.prod(shape)), np.float32)
    out[G["%s_%s_idx" % (tag, key)]] = G["%s_%s_val" % (tag, key)]
    return out.reshape(shape)


def check(tag, out, shape, masks=MASKS):
    n = int(np.prod(shape))
    for k in masks:
        want = np.unpackbits(G["%s_%s" % (tag, k)])[:n].astype(bool).reshape(shape)
        got = np.asarray(out[k].cpu() if torch.is_tensor(out[k]) else out[k]).astype(bool)
        assert np.array_equal(got, want), (tag, k, int((got != want).sum()))
    for k in ("general_cls_loss_mask_float", "general_reg_loss_mask_float"):
        got = out[k].cpu().numpy()
        assert np.array_equal(got, dense(tag, k, shape)), (tag, k)
    assert int(out["pos_all_num"]) == int(G["%s_pos_all_num" % tag])


def dropped_of(tag, M):
    return torch.from_numpy(np.unpackbits(G["%s_dropped" % tag])[:M].astype(bool))


@pytest.mark.parametrize("tag", sorted(CASES))
def test_oracle_options_vs_reference(tag):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import common
    cfg = option_cfg(**CASES[tag])
    O = occ_oracle.OccOracle(cfg)
    _, _, bd = golden_batch()
    M = bd["voxels"].shape[0]
    dropped = dropped_of(tag, M) if tag.startswith("drop") else None
    out = O.targets(bd, dropped=dropped)
    check(tag, out, (bd["gt_boxes"].shape[0], O.nz, O.ny, O.nx))
    if dropped is not None:
        assert int(dropped.sum()) == int(G["%s_n_dropped" % tag]) > 100
        assert out["voxels"].shape[0] == int(G["%s_n_voxels_out" % tag])
        assert np.array_equal(common.sha1(out["voxels"].numpy()), G["%s_voxels_sha1" % tag])
        want = np.unpackbits(G["%s_fore_voxel_drop_mask" % tag])[:out["fore_voxel_drop_mask"].numel()].astype(bool)
        assert np.array_equal(out["fore_voxel_drop_mask"].numpy().astype(bool).reshape(-1), want)
    if tag == "back_track":      # differs from NOTHING only on rays without any hit
        base = occ_oracle.OccOracle(load_cfg()).targets(bd)
        assert int((base["_occ_raw"] != out["_occ_raw"]).sum()) > 0
