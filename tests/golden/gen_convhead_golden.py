"""Golden vectors of the ROI head's pooling stage from the REFERENCE's own ConvHead
(btcdet/models/roi_heads/conv_head.py), executed here: its sparse layers over tests/golden/oracle_spconv.py (spconv := the C
oracle) and its compiled pointnet2_stack primitives (ball query, grouping: CUDA-only in the reference) served by the C oracle's
restatements (oracle/btc_oracle.c orc_ball_query / orc_group_points).

    python tests/golden/gen_convhead_golden.py        # writes tests/golden/convhead.npz

Inputs (common.convhead_inputs) and weights (common.init_by_name) are regenerated on the test side; stored are digests of
roi_conv_pool's output and of the head's cls / reg predictions, eval-mode and train-mode BatchNorm (dropout off: eval() for the
former, DP layers set to p = 0 for the latter)."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_env  # noqa: E402
import oracle_spconv  # noqa: E402

ref_env.install(oracle_spconv)

import ctypes  # noqa: E402

import torch  # noqa: E402

import common  # noqa: E402
from oracle import oracle as orc  # noqa: E402

# ---- the compiled pointnet2_stack module, served by the oracle (same argument lists as the reference's pybind wrappers)
_f = lambda t: t.numpy().ctypes.data_as(ctypes.POINTER(ctypes.c_float))
_i = lambda t: t.numpy().ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx):
    orc.lib().orc_ball_query(_f(new_xyz), _i(new_xyz_batch_cnt), _f(xyz), _i(xyz_batch_cnt), int(B), int(M), ctypes.c_float(-1.0), ctypes.c_float(float(radius)),
                             int(nsample), _i(idx))


def group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out):
    orc.lib().orc_group_points(_f(features), _i(features_batch_cnt), _i(idx), _i(idx_batch_cnt), int(B), int(M), int(C), int(nsample), _f(out))


stub = sys.modules["btcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda"]
stub.ball_query_wrapper, stub.group_points_wrapper = ball_query_wrapper, group_points_wrapper
torch.cuda.IntTensor = lambda *s: torch.zeros(*s, dtype=torch.int32)       # the reference allocates its outputs as torch.cuda.*Tensor
torch.cuda.FloatTensor = lambda *s: torch.zeros(*s, dtype=torch.float32)

from btcdet.models.roi_heads.conv_head import ConvHead  # noqa: E402


def main():
    cfg = ref_env.load_ref_cfg()
    inp = common.convhead_inputs()
    head = ConvHead(input_channels=128, model_cfg=cfg.MODEL.ROI_HEAD, num_class=1, det_voxel_size=[0.05, 0.05, 0.1],
                    point_cloud_range=np.array(cfg.DATA_CONFIG.POINT_CLOUD_RANGE, dtype=np.float32), num_rawpoint_features=4,
                    pre_conv_num_bev_features=None)
    common.init_by_name(head)
    for m in head.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    gold = {"n_rois": np.array(inp["rois"].shape[1]), "state_keys": np.array(sorted(head.state_dict().keys()))}
    for k in ("points", "occ_pnts", "xc_features", "rois"):
        gold["in_%s_sha1" % k] = common.sha1(inp[k])
    gold["in_xc_indices_sha1"] = common.sha1(inp["xc_indices"])
    for mode in ("eval", "train"):
        head.train(mode == "train")
        state = {k: v.clone() for k, v in head.state_dict().items()}
        xc = oracle_spconv.SparseConvTensor(torch.from_numpy(inp["xc_features"].copy()), torch.from_numpy(inp["xc_indices"].copy()), inp["xc_shape"], 2)
        bd = {"batch_size": 2, "rois": torch.from_numpy(inp["rois"].copy()), "points": torch.from_numpy(inp["points"].copy()),
              "occ_pnts": torch.from_numpy(inp["occ_pnts"].copy()), "added_occ_b_ind": torch.from_numpy(inp["added_occ_b_ind"].copy()),
              "multi_scale_3d_features": {"x_combine": xc}}
        with torch.no_grad():
            pooled, _ = head.roi_conv_pool(bd)
            shared = head.shared_fc_layer(pooled)
            cls = head.cls_layers(shared).transpose(1, 2).contiguous().squeeze(dim=1)
            reg = head.reg_layers(shared).transpose(1, 2).contiguous().squeeze(dim=1)
        head.load_state_dict(state)
        common.put_digest(gold, mode + "_pooled", pooled.numpy(), n=40000)
        # per-source slices of the pooled tensor (channel order of roi_conv_pool's concatenation: raw points 64, occupancy points 48,
        # x_combine 128)
        v = pooled.view(pooled.shape[0], -1, 27)     # (BN, C, 27)
        for name, sl in (("raw", slice(0, 64)), ("occ", slice(64, 112)), ("xc", slice(112, 240))):
            common.put_digest(gold, "%s_pooled_%s" % (mode, name), v[:, sl].contiguous().numpy(), n=20000)
        gold[mode + "_rcnn_cls"] = cls.numpy()
        gold[mode + "_rcnn_reg"] = reg.numpy()
        print(mode, "pooled", tuple(pooled.shape), "abs mean %.4f" % float(pooled.abs().mean()), "cls", tuple(cls.shape), "reg", tuple(reg.shape))
    out = os.path.join(HERE, "convhead.npz")
    np.savez_compressed(out, **gold)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024))


if __name__ == "__main__":
    main()
