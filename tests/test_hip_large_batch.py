"""batch x grid volumes beyond 2^31 cells (the 32-bit cell-key limit the round-1 review flagged: the KITTI detection grid
[40, 1600, 1408] fails from 24 scenes per batch on): voxelizer and rulebooks with 64-bit cell indices, against the oracle run
scene by scene (scenes are independent given the batch column)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

from btcdet_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_voxelizer_and_rulebooks_at_batch_28():
    from btcdet_amd.spconv import ops, utils
    B = 28                                                   # 28 x 40 x 1600 x 1408 = 2.5e9 cells > 2^31
    scenes = [synth.make_scene(500 + b, az_step=2.4) for b in range(B)]     # ~2 k points each
    pts = np.concatenate([s["points"] for s in scenes])
    offs = np.cumsum([0] + [s["points"].shape[0] for s in scenes]).astype(np.int32)
    gen = utils.VoxelGeneratorV2(synth.KITTI_DET_VOXEL, synth.KITTI_DET_RANGE, 5, 16000)
    v, c, n = gen.generate_batch(torch.from_numpy(pts).to(DEV), torch.from_numpy(offs).to(DEV))
    ogen = orc.VoxelGeneratorV2(synth.KITTI_DET_VOXEL, synth.KITTI_DET_RANGE, 5, 16000)
    ref_c, ref_v = [], []
    for b, s in enumerate(scenes):
        r = ogen.generate(s["points"])
        ref_c.append(np.pad(r["coordinates"], ((0, 0), (1, 0)), constant_values=b))
        ref_v.append(r["voxels"])
    ref_c, ref_v = np.concatenate(ref_c), np.concatenate(ref_v)
    np.testing.assert_array_equal(c.cpu().numpy(), ref_c)
    np.testing.assert_array_equal(v.cpu().numpy(), ref_v)
    shape = [41, 1600, 1408]
    for kw, mode in ((dict(ksize=3, stride=1, padding=1, subm=True), orc.MODE_SUBM), (dict(ksize=3, stride=2, padding=1, subm=False), orc.MODE_CONV)):
        rb = ops.build_rulebook(c, B, shape, kw["ksize"], kw["stride"], kw["padding"], 1, 0, kw["subm"], False)
        o_idx, o_out, o_in, _ = orc.rulebook(ref_c, shape, 3, kw["stride"], 0 if kw["subm"] else 1, 1, mode)
        np.testing.assert_array_equal(rb.out_indices.cpu().numpy(), o_idx)
        np.testing.assert_array_equal(rb.nbr_out.cpu().numpy(), o_out)
        np.testing.assert_array_equal(rb.nbr_in.cpu().numpy(), o_in)
