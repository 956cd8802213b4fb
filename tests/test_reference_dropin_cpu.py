"""Drop-in check against the reference's OWN backbone file (runs only where /root/reference is mounted, i.e. in the build
container -- never on the GPU box): /root/reference/btcdet/models/backbones_3d/spconv_backbone.py imports `spconv`; with
btcdet_amd.install_as_spconv() it imports THIS implementation, its classes construct on top of it, and their state_dict
(keys, shapes) is identical to the one of this repository's backbones, so reference checkpoints load key-for-key."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference/btcdet/models/backbones_3d/spconv_backbone.py"
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not mounted")


@pytest.fixture(scope="module")
def ref_mod():
    import btcdet_amd
    saved = {k: sys.modules.get(k) for k in ("spconv", "spconv.utils", "spconv.ops")}
    btcdet_amd.install_as_spconv()
    spec = importlib.util.spec_from_file_location("_ref_spconv_backbone", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    yield mod
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def _cfgs():
    from btcdet_amd.config import load_cfg
    cfg = load_cfg()
    return cfg, cfg.MODEL.OCC.BACKBONE_3D, cfg.MODEL.BACKBONE_3D


def _same_state(mine, ref):
    a, b = mine.state_dict(), ref.state_dict()
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert tuple(a[k].shape) == tuple(b[k].shape), k
    mine.load_state_dict(b, strict=True)


def test_reference_module_uses_this_spconv(ref_mod):
    import btcdet_amd.spconv as sp
    assert ref_mod.spconv is sp
    blk = ref_mod.post_act_block(16, 32, 3, indice_key="k", stride=2, padding=1, conv_type="spconv",
                                 norm_fn=lambda c: torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01))
    assert isinstance(blk, sp.SparseSequential) and isinstance(blk[0], sp.SparseConv3d)
    assert tuple(blk[0].weight.shape) == (3, 3, 3, 16, 32)


def test_occupancy_backbone_state_dict_matches_reference(ref_mod):
    from btcdet_amd import backbones_3d
    cfg, occ_cfg, _ = _cfgs()
    grid = np.array([209, 157, 9])
    kw = dict(model_cfg=occ_cfg, input_channels=4, grid_size=grid, voxel_size=[0.32, 0.5184, 0.36],
              point_cloud_range=cfg.DATA_CONFIG.POINT_CLOUD_RANGE, original_num_rawpoint_features=4)
    mine = backbones_3d.VoxelBackBoneDeconv(**kw)
    ref = ref_mod.VoxelBackBoneDeconv(**kw)
    _same_state(mine, ref)
    assert mine.num_point_features == ref.num_point_features
    assert list(mine.sparse_shape) == list(ref.sparse_shape)


def test_detection_backbone_state_dict_matches_reference(ref_mod):
    from btcdet_amd import backbones_3d
    cfg, _, det_cfg = _cfgs()
    grid = np.array([1408, 1600, 40])
    kw = dict(model_cfg=det_cfg, input_channels=6, grid_size=grid, voxel_size=[0.05, 0.05, 0.1],
              point_cloud_range=cfg.DATA_CONFIG.POINT_CLOUD_RANGE, original_num_rawpoint_features=4)
    mine = backbones_3d.VoxelBackBone8xOcc(**kw)
    ref = ref_mod.VoxelBackBone8xOcc(**kw)
    _same_state(mine, ref)
    assert mine.num_point_features == ref.num_point_features
    assert list(mine.sparse_shape) == list(ref.sparse_shape)


@pytest.mark.parametrize("name,cin,grid", [("VoxelBackBoneDeconvRes", 4, [209, 157, 9]), ("VoxelBackBoneInverseRes", 4, [209, 157, 9]),
                                           ("VoxelResBackBone8x", 4, [1408, 1600, 40])])
def test_unconfigured_backbone_variants_match_reference(ref_mod, name, cin, grid, capsys):
    """the variants reachable through BACKBONE_3D.NAME"""
    from btcdet_amd import backbones_3d
    cfg, occ_cfg, _ = _cfgs()
    kw = dict(model_cfg=occ_cfg, input_channels=cin, grid_size=np.array(grid))
    mine = backbones_3d.__all__[name](**kw)
    ref = getattr(ref_mod, name)(**kw)
    _same_state(mine, ref)
    assert mine.num_point_features == ref.num_point_features and list(mine.sparse_shape) == list(ref.sparse_shape)


def test_sparse_basic_block_matches_reference(ref_mod):
    from functools import partial
    from btcdet_amd import backbones_3d
    norm_fn = partial(torch.nn.BatchNorm1d, eps=1e-3, momentum=0.01)
    _same_state(backbones_3d.SparseBasicBlock(32, 32, norm_fn=norm_fn, indice_key="res2"),
                ref_mod.SparseBasicBlock(32, 32, norm_fn=norm_fn, indice_key="res2"))


HOT = ("occ_modules.", "det_modules.vfe", "det_modules.backbone_3d", "det_modules.map_to_bev_module", "global_step")


def _hot_path_state():
    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    return {k: list(v.shape) for k, v in BtcHotPath(load_cfg(), device="cpu").state_dict().items()}


def test_reference_btcnet_builds_on_this_spconv_and_hot_path_state_dict_matches():
    """the reference's OWN BtcNet (detector template, registries, yaml) constructs with `spconv` = btcdet_amd.spconv; the
    hot-path part of its state_dict (occ_modules.*, det_modules.{vfe, backbone_3d, map_to_bev_module}) equals BtcHotPath's
    key for key, shape for shape, in order -- so a reference checkpoint loads into the hot path and vice versa"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "golden", "ref_build_state.py")], capture_output=True, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    res = json.loads(out.stdout.decode().strip().splitlines()[-1])
    # the reference's iou3d_nms_utils.py imported with its compiled module resolved to btcdet_amd.iou3d_nms.iou3d_nms_cuda
    assert res["iou3d_nms_utils_bound_to_btcdet_amd"] is True
    ref = res["keys"]
    assert len(ref) > 300  # the whole detector built, incl. the out-of-scope 2-D backbone and heads
    hot = {k: v for k, v in ref.items() if k.startswith(HOT)}
    mine = _hot_path_state()
    assert list(hot.keys()) == list(mine.keys())
    assert hot == mine
    committed = json.load(open(os.path.join(root, "tests", "golden", "ref_state_keys.json")))["keys"]
    assert committed == ref  # the fixture used where the reference is not mounted is current


def test_reference_pointnet2_stack_binds_the_stand_in():
    """the reference's pointnet2_utils.py / pointnet2_modules.py import `pointnet2_stack_cuda`; with
    btcdet_amd.pointnet2_stack.install_as_pointnet2_stack_cuda() they import this implementation's entry points, and the
    reference's StackSAModuleMSG has the same parameters as this repository's"""
    import types
    from btcdet_amd import pointnet2_stack as p2
    pkg = "btcdet.ops.pointnet2.pointnet2_stack"
    root = "/root/reference/btcdet/ops/pointnet2/pointnet2_stack"
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "btcdet" or k.startswith("btcdet.")}
    try:
        parts = pkg.split(".")
        for i in range(1, len(parts) + 1):
            name = ".".join(parts[:i])
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
        stand_in = p2.install_as_pointnet2_stack_cuda(pkg)
        mods = {}
        for fname in ("pointnet2_utils", "pointnet2_modules"):
            spec = importlib.util.spec_from_file_location(pkg + "." + fname, os.path.join(root, fname + ".py"))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[pkg + "." + fname] = mod
            setattr(sys.modules[pkg], fname, mod)
            spec.loader.exec_module(mod)
            mods[fname] = mod
        assert mods["pointnet2_utils"].pointnet2 is stand_in
        for name in ("ball_query_wrapper", "shell_query_wrapper", "group_points_wrapper", "group_points_grad_wrapper",
                     "furthest_point_sampling_wrapper", "three_nn_wrapper", "three_interpolate_wrapper", "three_interpolate_grad_wrapper"):
            assert callable(getattr(stand_in, name))
        kw = dict(radii=[0.8, 1.6], nsamples=[16, 16], use_xyz=True, pool_method="max_pool")
        ref = mods["pointnet2_modules"].StackSAModuleMSG(mlps=[[1, 16, 16], [1, 16, 16]], **kw)
        mine = p2.StackSAModuleMSG(mlps=[[1, 16, 16], [1, 16, 16]], **kw)
        _same_state(mine, ref)
        ref_fp = mods["pointnet2_modules"].StackPointnetFPModule(mlp=[32, 16])
        _same_state(p2.StackPointnetFPModule(mlp=[32, 16]), ref_fp)
    finally:
        for k in [k for k in sys.modules if k == "btcdet" or k.startswith("btcdet.")]:
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
