"""Golden vectors for the backbone variants the reference reaches through BACKBONE_3D.NAME, produced by
the REFERENCE's own classes executed here over tests/golden/oracle_spconv.py (spconv := the C oracle):

    VoxelBackBoneDeconvRes   spconv_backbone.py:226-381   (residual blocks + lateral merges through `combine`)
    VoxelBackBoneInverseRes  spconv_backbone.py:385-527   (the same with SparseInverseConv3d decoders)
    VoxelResBackBone8x       spconv_backbone.py:531-627   (the residual 8x detection backbone)

    python tests/golden/gen_variants_golden.py        # writes tests/golden/variants.npz

Inputs (common.variant_inputs: synthetic KITTI scenes through the oracle's voxelizer) and weights (common.init_by_name) are
regenerated on the test side; stored are index digests (SHA-1) and float digests of every output, train- and eval-mode
BatchNorm.  The reference files are imported, never copied; nothing here travels to the GPU box but the .npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_env  # noqa: E402
import oracle_spconv  # noqa: E402

ref_env.install(oracle_spconv)

import torch  # noqa: E402

import common  # noqa: E402
from btcdet.models.backbones_3d import spconv_backbone as ref_bb  # noqa: E402


def run(name, which, gold):
    cfg = ref_env.load_ref_cfg()
    feats, coords, grid, B = common.variant_inputs(which)
    model_cfg = cfg.MODEL.OCC.BACKBONE_3D if which == "occ" else cfg.MODEL.BACKBONE_3D
    net = getattr(ref_bb, name)(model_cfg=model_cfg, input_channels=4, grid_size=np.array(grid))
    common.init_by_name(net)
    gold["%s_n_in" % name] = np.array(coords.shape[0])
    gold["%s_in_coords_sha1" % name] = common.sha1(coords)
    gold["%s_in_feats_sha1" % name] = common.sha1(feats)
    for mode in ("train", "eval"):
        net.train(mode == "train")
        state = {k: v.clone() for k, v in net.state_dict().items()}
        with torch.no_grad():
            bd = net({"voxel_features": torch.from_numpy(feats.copy()), "voxel_coords": torch.from_numpy(coords.copy()).float(), "batch_size": B})
        net.load_state_dict(state)
        outs = {"out": bd["encoded_spconv_tensor"]}
        for k, v in (bd.get("multi_scale_3d_features") or {}).items():
            outs[k] = v
        for key, x in outs.items():
            p = "%s_%s_%s_" % (name, mode, key)
            gold[p + "indices_sha1"] = common.sha1(x.indices.numpy().astype(np.int32))
            gold[p + "n"] = np.array(x.features.shape[0])
            gold[p + "shape"] = np.array([int(v) for v in x.spatial_shape])
            common.put_digest(gold, p + "features", x.features.numpy())
        print("  %s %s: %s" % (name, mode, {k: tuple(v.features.shape) for k, v in outs.items()}))


if __name__ == "__main__":
    gold = {}
    for name, which in (("VoxelBackBoneDeconvRes", "occ"), ("VoxelBackBoneInverseRes", "occ"), ("VoxelResBackBone8x", "det")):
        print("==", name)
        run(name, which, gold)
    out = os.path.join(HERE, "variants.npz")
    np.savez_compressed(out, **gold)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024))
