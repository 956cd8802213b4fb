"""CPU-side checks: the C-ABI library loads and exports every symbol include/btcdet_hip.h declares, the ctypes
signatures cover them all, and the pure-host entry points behave (no compute call is made without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "btcdet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(btc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from btcdet_amd import _lib
    L = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libbtcdet_hip.so does not export %s" % n
    assert sorted(_lib.EXPORTED_SYMBOLS) == names, "ctypes signature table and header disagree"
    assert L.btc_version() >= 1


def test_host_only_entry_points():
    from btcdet_amd import _lib
    from oracle import oracle as orc
    L = _lib.lib()
    i3, p = _lib.i3, _lib.i3p
    cases = [([41, 1600, 1408], 3, 2, 1, 1, 1, [21, 800, 704]), ([11, 400, 352], 3, 2, (0, 1, 1), 1, 1, [5, 200, 176]),
             ([5, 200, 176], (3, 1, 1), (2, 1, 1), 0, 1, 1, [2, 200, 176]), ([3, 40, 53], 3, 2, 1, 1, 2, [5, 79, 105]),
             ([9, 157, 209], 3, 1, 1, 1, 0, [9, 157, 209])]
    for shp, k, s, pad, d, mode, want in cases:
        out = np.zeros(3, np.int32)
        assert L.btc_out_shape(p(i3(shp)), p(i3(k)), p(i3(s)), p(i3(pad)), p(i3(d)), p(i3(0)), mode, p(out)) == 0
        assert out.tolist() == want == orc.out_shape(shp, k, s, pad, d, mode).tolist()
    assert L.btc_out_shape(p(i3(1)), p(i3(1)), p(i3(1)), p(i3(0)), p(i3(1)), p(i3(0)), 7, p(np.zeros(3, np.int32))) == _lib.ci(-1).value
    assert b"bad mode" in L.btc_last_error()
    # workspace helpers are monotone and non-trivial
    assert L.btc_voxelize_ws_bytes(60000, 2, 12) > L.btc_voxelize_ws_bytes(1000, 2, 12) > 0
    assert L.btc_rulebook_subm_ws_bytes(40000) >= 2 * 4 * 65536
    assert L.btc_conv_wgrad_ws_bytes(15000, 27, 256, 128, -1) >= 27 * 256 * 128 * 4
    cfg = _lib.BtcOccConfig()
    cfg.batch, cfg.grid[:], cfg.sphere_grid[:] = 2, [209, 157, 9], [214, 157, 49]
    assert L.btc_occ_targets_ws_bytes(ctypes.byref(cfg)) > 2 * 209 * 157 * 9 * (9 * 4 + 3 * 4)


def test_environment_switches_of_the_loader(monkeypatch):
    """BTC_TUNE="key=value,..." reaches btc_tune_set when the library is loaded (a bad entry is an error, not ignored);
    BTC_TUNE_APPLY_DEBUG -- the one key that changes results -- is refused unless BTC_ALLOW_WRONG_RESULTS is set;
    BTC_FASTPATH=0 keeps the compiled torch binding out (the ctypes route of INTEGRATION.md section 2 stays)"""
    from btcdet_amd import _lib
    L0 = _lib.lib()
    try:
        monkeypatch.setattr(_lib, "_lib", None)
        monkeypatch.setenv("BTC_TUNE", "14=1,20=2")
        L = _lib.lib()
        assert L.btc_tune_value(14) == 1 and L.btc_tune_value(20) == 2 and L.btc_tune_value(15) == 0
        monkeypatch.setattr(_lib, "_lib", None)
        monkeypatch.setenv("BTC_TUNE", "9999=1")
        with pytest.raises(_lib.BtcHipError):
            _lib.lib()
        monkeypatch.delenv("BTC_TUNE")
        monkeypatch.delenv("BTC_ALLOW_WRONG_RESULTS", raising=False)
        assert L.btc_tune_set(3, 1) != 0 and L.btc_tune_value(3) == 0
        assert L.btc_tune_set(3, 0) == 0
        monkeypatch.setenv("BTC_ALLOW_WRONG_RESULTS", "1")
        assert L.btc_tune_set(3, 1) == 0 and L.btc_tune_value(3) == 1
    finally:
        for k in (3, 14, 20):
            L0.btc_tune_set(k, 0)
        monkeypatch.setattr(_lib, "_lib", L0)
    monkeypatch.setattr(_lib, "_fast", False)
    monkeypatch.setenv("BTC_FASTPATH", "0")
    assert _lib.fast() is None
    monkeypatch.setattr(_lib, "_fast", False)
    monkeypatch.setenv("BTC_FASTPATH", "1")
    assert _lib.fast() is not None, "the compiled binding is not built"


def test_no_fallback_when_library_missing(monkeypatch):
    from btcdet_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libbtcdet_hip.so")
    with pytest.raises(_lib.BtcHipError):
        _lib.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "btcdet_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, f)


def test_state_dict_keys_match_reference_layout():
    """parameter names / shapes follow the reference's module nesting so its checkpoints load key-for-key
    (spconv_backbone.py:106-128,656-767; occ_head_3D.py:25-31; detector3d_template.py:40-41)"""
    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    sd = BtcHotPath(load_cfg(), device="cpu").state_dict()
    want = {
        "occ_modules.backbone_3d.conv1.0.0.weight": (3, 3, 3, 4, 16), "occ_modules.backbone_3d.conv2.1.0.weight": (3, 3, 3, 32, 32),
        "occ_modules.backbone_3d.deconv5.0.0.weight": (3, 3, 3, 32, 32), "occ_modules.backbone_3d.deconv4.0.1.running_var": (32,),
        "occ_modules.occ_dense_head.conv_cls.0.weight": (3, 3, 3, 32, 2), "occ_modules.occ_dense_head.conv_cls.0.bias": (2,),
        "occ_modules.occ_dense_head.conv_res.0.weight": (3, 3, 3, 32, 3), "occ_modules.occ_targets.fix_conv_2dzy.weight": (1, 1, 3, 3),
        "det_modules.backbone_3d.conv1.0.weight": (3, 3, 3, 6, 16), "det_modules.backbone_3d.conv1_combine.0.0.weight": (3, 3, 3, 16, 16),
        "det_modules.backbone_3d.conv2_combine.0.0.weight": (3, 3, 3, 34, 32), "det_modules.backbone_3d.conv4.0.0.weight": (3, 3, 3, 64, 64),
        "det_modules.backbone_3d.conv_out.0.weight": (3, 1, 1, 64, 128), "det_modules.backbone_3d.down2.1.0.weight": (3, 3, 3, 32, 64),
        "det_modules.backbone_3d.squeezeBev.0.0.weight": (2, 1, 1, 128, 64), "det_modules.backbone_3d.down_combine.0.0.weight": (3, 3, 3, 256, 128),
        "det_modules.backbone_3d.down_combine.1.0.weight": (3, 3, 3, 128, 128), "global_step": (1,),
    }
    for k, shape in want.items():
        assert k in sd and tuple(sd[k].shape) == shape, k
    assert not any("occ_conv2" in k for k in sd)  # the max-pool branch has no parameters


def test_hot_path_state_dict_matches_reference_fixture():
    """tests/golden/ref_state_keys.json = state_dict keys / shapes of the reference's own BtcNet build (generated by
    tests/golden/ref_build_state.py where the reference is mounted); the hot-path subset must equal BtcHotPath's"""
    import json
    import os
    from btcdet_amd.btc_path import BtcHotPath
    from btcdet_amd.config import load_cfg
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = json.load(open(os.path.join(root, "tests", "golden", "ref_state_keys.json")))["keys"]
    hot = {k: v for k, v in ref.items() if k.startswith(("occ_modules.", "det_modules.vfe", "det_modules.backbone_3d",
                                                         "det_modules.map_to_bev_module", "global_step"))}
    mine = {k: list(v.shape) for k, v in BtcHotPath(load_cfg(), device="cpu").state_dict().items()}
    assert list(hot.keys()) == list(mine.keys()) and hot == mine
