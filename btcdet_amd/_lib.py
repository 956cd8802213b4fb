"""ctypes binding of libbtcdet_hip.so (C ABI in include/btcdet_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails this module raises.
torch is used only to own device memory and to name the current HIP stream.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbtcdet_hip.so")
_lib = None

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_f32p = ctypes.POINTER(ctypes.c_float)
vp = ctypes.c_void_p
ci = ctypes.c_int
sz = ctypes.c_size_t

MODE_SUBM, MODE_CONV, MODE_TRANSPOSE = 0, 1, 2


class BtcHipError(RuntimeError):
    pass


class BtcOccConfig(ctypes.Structure):
    """struct BtcOccConfig of include/btcdet_hip.h"""
    _fields_ = [("batch", ctypes.c_int32), ("grid", ctypes.c_int32 * 3), ("sphere_grid", ctypes.c_int32 * 3),
                ("dist_kern", ctypes.c_int32 * 3), ("concede_x", ctypes.c_int32), ("empt_sur_thresh", ctypes.c_int32),
                ("max_boxes", ctypes.c_int32), ("use_box_weight", ctypes.c_int32),
                ("occ_range", ctypes.c_float * 6), ("occ_voxel", ctypes.c_float * 3),
                ("sphere_range", ctypes.c_float * 6), ("sphere_voxel", ctypes.c_float * 3),
                ("det_zmin", ctypes.c_float), ("det_zmax", ctypes.c_float),
                ("w_fore_cls", ctypes.c_float), ("w_mirr_cls", ctypes.c_float), ("w_bm_cls", ctypes.c_float),
                ("w_neg_cls", ctypes.c_float), ("w_fore_res", ctypes.c_float), ("w_mirr_res", ctypes.c_float),
                ("w_bm_res", ctypes.c_float), ("box_weight", ctypes.c_float), ("backproject_lut", ctypes.c_void_p),
                ("reverse_vis", ctypes.c_int32), ("vis_half", ctypes.c_int32)]


OCC_BUFFER_FIELDS = ["vcc_mask", "voxelwise_mask", "bm_voxelwise_mask", "occ_voxelwise_mask", "fore_voxelwise_mask",
                     "pos_mask", "general_cls_loss_mask", "occ_fore_cls_mask", "occ_mirr_cls_mask", "occ_bm_cls_mask",
                     "general_reg_loss_mask", "forebox_label", "general_cls_loss_mask_float",
                     "general_reg_loss_mask_float", "res_mtrx", "pos_all_num"]


class BtcOccBuffers(ctypes.Structure):
    """struct BtcOccBuffers of include/btcdet_hip.h (all device pointers)"""
    _fields_ = [(k, ctypes.c_void_p) for k in OCC_BUFFER_FIELDS]


class BtcChainLayer(ctypes.Structure):
    """struct BtcChainLayer of include/btcdet_hip.h"""
    _fields_ = [("kind", ctypes.c_int32), ("ref", ctypes.c_int32), ("mode", ctypes.c_int32), ("in_shape", ctypes.c_int32 * 3),
                ("out_shape", ctypes.c_int32 * 3), ("k", ctypes.c_int32 * 3), ("s", ctypes.c_int32 * 3), ("p", ctypes.c_int32 * 3),
                ("d", ctypes.c_int32 * 3)]


class BtcPovConfig(ctypes.Structure):
    """struct BtcPovConfig of include/btcdet_hip.h"""
    _fields_ = [("batch", ctypes.c_int32), ("max_k", ctypes.c_int32), ("occ_grid", ctypes.c_int32 * 3), ("det_grid", ctypes.c_int32 * 3),
                ("occ_origin", ctypes.c_float * 3), ("occ_voxel", ctypes.c_float * 3), ("det_origin", ctypes.c_float * 3),
                ("det_voxel", ctypes.c_float * 3), ("occ_thresh", ctypes.c_float), ("inten", ctypes.c_float), ("code_dim", ctypes.c_int32)]


_SIGS = {
    # name: (restype, argtypes)
    "btc_last_error": (ctypes.c_char_p, []),
    "btc_version": (ci, []),
    "btc_tune_set": (ci, [ci, ci]),
    "btc_tune_value": (ci, [ci]),
    "btc_mean_vfe": (ci, [vp, vp, ci, ci, ci, ci, vp, vp]),
    "btc_occ_vfe": (ci, [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp]),
    "btc_boxes_pairwise_bev": (ci, [vp, ci, vp, ci, ci, vp, vp]),
    "btc_nms_ws_bytes": (sz, [ci]),
    "btc_nms": (ci, [vp, ci, ctypes.c_float, ci, vp, vp, vp, sz, vp]),
    "btc_nms_topk_ws_bytes": (sz, [ci, ci, ci]),
    "btc_trilinear_corners": (ci, [vp, ctypes.c_longlong, ctypes.c_longlong, c_f32p, c_f32p, c_f32p, c_i32p, ci, vp, vp, vp, vp, vp, vp]),
    "btc_trilinear_gather": (ci, [vp, ci, ctypes.c_longlong, vp, vp, vp, vp]),
    "btc_trilinear_scatter": (ci, [vp, ci, vp, vp, vp, ci, vp, vp]),
    "btc_nms_topk": (ci, [vp, ci, ci, ctypes.c_float, ci, ci, vp, vp, vp, sz, vp]),
    "btc_voxelize_ws_bytes": (sz, [ci, ci, ci]),
    "btc_voxelize": (ci, [vp, ci, ci, ci, ci, ci, vp, ci, c_f32p, c_f32p, c_i32p, ci, ci, vp, vp, vp, vp, vp, sz, vp]),
    "btc_cart_to_occ_coords": (ci, [vp, vp, ci, ci, ci, vp]),
    "btc_voxel_shift_col": (ci, [vp, vp, ci, ci, ci, ci, vp, ctypes.c_float, vp]),
    "btc_range_mask_ws_bytes": (sz, [ci]),
    "btc_range_mask_compact": (ci, [vp, vp, ci, ci, ci, vp, ci, c_f32p, vp, vp, vp, vp, vp, sz, vp]),
    "btc_gather_rows": (ci, [vp, vp, ci, ci, ci, vp, vp, vp]),
    "btc_out_shape": (ci, [c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, ci, c_i32p]),
    "btc_rulebook_subm_ws_bytes": (sz, [ci]),
    "btc_rulebook_subm": (ci, [vp, ci, ci, c_i32p, c_i32p, c_i32p, vp, vp, vp, sz, vp]),
    "btc_rulebook_conv_ws_bytes": (sz, [ci, c_i32p]),
    "btc_rulebook_conv_count": (ci, [vp, ci, ci, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, ci, vp, vp, sz, vp]),
    "btc_rulebook_conv_fill": (ci, [vp, ci, ci, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, ci, ci, vp, vp, vp, vp, sz, vp]),
    "btc_pairs_from_nbr_ws_bytes": (sz, [ci, ci]),
    "btc_pairs_from_nbr": (ci, [vp, ci, ci, ci, vp, vp, vp, sz, vp]),
    "btc_chain_ws_bytes": (sz, [vp, ci, ci, ci]),
    "btc_chain_caps": (ci, [vp, ci, ci, ci, vp]),
    "btc_chain_levels": (ci, [vp, ci, ci, vp, ci, vp, vp, vp, vp, sz, vp]),
    "btc_chain_maps": (ci, [vp, ci, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "btc_conv_fwd": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp]),
    "btc_conv_dgrad": (ci, [vp, vp, vp, ci, ci, ci, ci, vp, vp]),
    "btc_conv_wgrad_ws_bytes": (sz, [ci, ci, ci, ci, ci]),
    "btc_conv_wgrad": (ci, [vp, vp, vp, ci, vp, ci, ci, ci, ci, vp, vp, sz, vp]),
    "btc_conv_fwd_bf16": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp]),
    "btc_conv_dgrad_bf16": (ci, [vp, vp, vp, ci, ci, ci, ci, vp, vp]),
    "btc_conv_wgrad_bf16": (ci, [vp, vp, vp, ci, vp, ci, ci, ci, ci, vp, vp, sz, vp]),
    "btc_conv_bf16w_supported": (ci, [ci, ci, ci]),
    "btc_weights_to_bf16": (ci, [vp, ci, ci, ci, vp, vp, vp]),
    "btc_conv_split_supported": (ci, [ci, ci, ci]),
    "btc_set_scratch": (ci, [vp, vp, sz]),
    "btc_conv_split_wanted": (ci, [ci, ci, ci, ci]),
    "btc_weights_split3": (ci, [vp, ci, ci, ci, vp, vp, vp]),
    "btc_weights_split3_multi": (ci, [vp, vp, vp, c_i32p, c_i32p, c_i32p, ci, vp]),
    "btc_weights_to_bf16_multi": (ci, [vp, vp, vp, c_i32p, c_i32p, c_i32p, ci, vp]),
    "btc_conv_fwd_bf16w": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp]),
    "btc_conv_dgrad_bf16w": (ci, [vp, vp, vp, ci, ci, ci, ci, vp, vp]),
    "btc_row_orders": (ci, [vp, c_i32p, c_i32p, ci, vp, vp]),
    "btc_row_orders_keyed": (ci, [vp, vp, c_i32p, c_i32p, ci, vp, vp]),
    "btc_conv_apply_ordered": (ci, [ci, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp]),
    "btc_conv_apply_src": (ci, [ci, ci, vp, ctypes.c_longlong, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp]),
    "btc_conv_wgrad_ordered": (ci, [ci, vp, vp, vp, ci, vp, ci, vp, vp, ci, ci, ci, vp, vp, sz, vp]),
    "btc_conv_wgrad_slabs": (ci, [ci, vp, vp, vp, ci, vp, ci, vp, vp, ci, ci, ci, vp, vp, sz, vp, vp]),
    "btc_wgrad_reduce_multi": (ci, [vp, vp, vp, vp, ci, vp]),
    "btc_adam_group_ws_bytes": (sz, [ci]),
    "btc_adam_max_segments": (ci, []),
    "btc_grads_pack": (ci, [vp, ci, vp, vp, vp, vp, c_i32p, vp, vp]),
    "btc_adam_group_step": (ci, [vp, ci, vp, vp, vp, vp, c_i32p, ci, vp, vp, vp, ctypes.c_longlong, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                 ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, sz, vp]),
    "btc_maxpool_fwd": (ci, [vp, vp, ci, ci, ci, vp, vp]),
    "btc_maxpool_bwd": (ci, [vp, vp, vp, vp, ci, ci, ci, vp, vp]),
    "btc_dense_fwd": (ci, [vp, vp, ci, ci, c_i32p, vp, vp]),
    "btc_dense_bwd": (ci, [vp, vp, ci, ci, c_i32p, vp, vp]),
    "btc_revoxelize_ws_bytes": (sz, [ci, ci, c_i32p]),
    "btc_revoxelize_count": (ci, [vp, ci, ci, c_i32p, vp, vp, vp, sz, vp]),
    "btc_revoxelize_fill": (ci, [vp, vp, ci, ci, ci, c_i32p, ci, ci, vp, vp, vp, vp, sz, vp]),
    "btc_pass_occ_vox_ws_bytes": (sz, [ctypes.POINTER(BtcPovConfig), ci, ci]),
    "btc_pass_occ_vox_count": (ci, [ctypes.POINTER(BtcPovConfig), vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, sz, vp]),
    "btc_pass_occ_vox_fill": (ci, [ctypes.POINTER(BtcPovConfig), vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, sz, vp]),
    "btc_pass_occ_vox_fill_i32": (ci, [ctypes.POINTER(BtcPovConfig), vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "btc_occ_loss_ws_bytes": (sz, []),
    "btc_occ_loss_fwd": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci, ctypes.c_longlong, ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, vp, vp,
                              sz, vp]),
    "btc_occ_loss_bwd": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci, ctypes.c_longlong, ctypes.c_float, vp, vp, vp, vp, vp]),
    "btc_occ_loss_fwd_total": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci, ctypes.c_longlong, ctypes.c_float, ctypes.c_float, ctypes.c_float, vp, vp,
                                    vp, sz, vp]),
    "btc_occ_loss_bwd_total": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci, ctypes.c_longlong, ctypes.c_float, vp, vp, vp, vp, vp]),
    "btc_dense_split_fwd": (ci, [vp, vp, ci, ci, ci, vp, vp, vp, vp]),
    "btc_occ_prob": (ci, [vp, vp, ci, ctypes.c_longlong, vp, vp]),
    "btc_dense_split_bwd": (ci, [vp, vp, vp, ci, ci, ci, vp, vp, vp]),
    "btc_cat_pad_fwd": (ci, [vp, ci, vp, ci, ctypes.c_longlong, ci, ci, vp, vp]),
    "btc_cat_pad_bwd": (ci, [vp, ci, ctypes.c_longlong, ci, vp, ci, vp, ci, vp]),
    "btc_sumsq2_ws_bytes": (sz, []),
    "btc_sumsq2_fwd": (ci, [vp, ctypes.c_longlong, ci, ctypes.c_double, vp, ctypes.c_longlong, ci, ctypes.c_double, vp, vp, sz, vp]),
    "btc_sumsq2_bwd": (ci, [vp, ctypes.c_longlong, ci, ctypes.c_float, vp, vp, ctypes.c_longlong, ci, ctypes.c_float, vp, vp, vp]),
    "btc_bn_ws_bytes": (sz, [ci]),
    "btc_bn_relu_fwd": (ci, [vp, ci, ci, vp, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, ci, ci, vp, vp, vp, vp, sz, vp]),
    "btc_spin": (ci, [ci, vp]),
    "btc_bn_relu_bwd": (ci, [vp, vp, vp, ci, ci, vp, vp, vp, ci, ci, vp, vp, vp, vp, sz, vp]),
    "btc_bn_relu_fwd_bf16": (ci, [vp, ci, ci, vp, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, ci, ci, vp, vp, vp, vp, sz, vp]),
    "btc_bn_relu_bwd_bf16": (ci, [vp, vp, vp, ci, ci, vp, vp, vp, ci, ci, vp, vp, vp, vp, sz, vp]),
    "btc_ball_query": (ci, [vp, vp, vp, vp, ci, ci, ctypes.c_float, ctypes.c_float, ci, vp, vp]),
    "btc_group_points": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp]),
    "btc_group_points_grad": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp]),
    "btc_furthest_point_sampling": (ci, [vp, ci, ci, ci, vp, vp, vp]),
    "btc_three_nn": (ci, [vp, vp, vp, vp, ci, ci, vp, vp, vp]),
    "btc_three_interpolate": (ci, [vp, vp, vp, ci, ci, vp, vp]),
    "btc_three_interpolate_grad": (ci, [vp, vp, vp, ci, ci, ci, vp, vp]),
    "btc_bn_fuse_ws_bytes": (sz, []),
    "btc_conv_bn_relu_fwd": (ci, [ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, ci, vp, vp, vp, vp, sz, vp, vp]),
    "btc_conv_bn_relu_fwd_src": (ci, [ci, vp, ctypes.c_longlong, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, ci, vp, vp, vp, vp, sz, vp, vp]),
    "btc_col_sum": (ci, [vp, ci, ci, vp, vp, sz, vp]),
    "btc_col_sum_bf16": (ci, [vp, ci, ci, vp, vp, sz, vp]),
    "btc_occ_targets_ws_bytes": (sz, [ctypes.POINTER(BtcOccConfig)]),
    "btc_occ_backproject_lut": (ci, [ctypes.POINTER(BtcOccConfig), vp, vp]),
    "btc_occ_targets": (ci, [ctypes.POINTER(BtcOccConfig), vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, ci, vp, vp,
                             ctypes.POINTER(BtcOccBuffers), vp, sz, vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGS.keys())


def lib():
    """Load the HIP library; raise loudly if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BtcHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C btcdet_amd/csrc` (there is no CPU fallback for the hot path)")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        # BTC_TUNE="key=value,key=value": tuning keys of include/btcdet_hip.h (BTC_TUNE_*) for A/B runs
        for kv in filter(None, os.environ.get("BTC_TUNE", "").split(",")):
            k, v = kv.split("=")
            if L.btc_tune_set(int(k), int(v)) != 0:
                raise BtcHipError("BTC_TUNE: bad entry %r" % kv)
        _lib = L
    return _lib


_fast = False


def fast():
    """the compiled PyTorch binding (btcdet_amd/_btcfast*.so, csrc/binding.cpp) or None when it has not been built or
    BTC_FASTPATH=0; the ctypes route above computes the same thing (it is the binding INTEGRATION.md documents)"""
    global _fast
    if _fast is False:
        _fast = None
        if os.environ.get("BTC_FASTPATH", "1") != "0":
            try:
                lib()  # load libbtcdet_hip.so first so that both bindings share one instance
                from . import _btcfast
                if _btcfast.abi_version() == lib().btc_version():
                    _fast = _btcfast
            except ImportError:
                _fast = None
    return _fast


def check(rc, what):
    if rc != 0:
        msg = lib().btc_last_error().decode("utf-8", "replace")
        raise BtcHipError(f"{what} failed (code {rc}): {msg}")


_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


def stream_ptr():
    """the current HIP stream of the current device as an integer handle (raw query: ~10x cheaper than
    torch.cuda.current_stream()); ctypes converts it to the void* the ABI takes"""
    return _raw_stream(_cur_device())


def ptr(t):
    """device address of a contiguous GPU tensor (None -> NULL).  Plain int / None: every entry point declares c_void_p
    argtypes, and this is called ~600 times per step on the launch-rate-bound path."""
    if t is None:
        return None
    if not (t.is_cuda and t.is_contiguous()):
        raise BtcHipError("expected a contiguous tensor on the GPU (HIP) device")
    return t.data_ptr()


_I3_CACHE = {}


def i3(v):
    """int / list / tuple / ndarray -> int32[3] array (accepts the reference's int|list|tuple kernel args); small-value
    triples are interned so that their ctypes pointers are built once"""
    if isinstance(v, (int, np.integer)):
        key = (int(v),) * 3
    else:
        key = tuple(int(x) for x in np.asarray(v).reshape(-1))
        if len(key) != 3:
            raise BtcHipError(f"expected 3 values, got {v!r}")
    a = _I3_CACHE.get(key)
    if a is None:
        a = np.ascontiguousarray(np.array(key, dtype=np.int32))
        a.setflags(write=False)
        a_ptr = a.ctypes.data_as(c_i32p)
        _I3_CACHE[key] = a
        _I3P_CACHE[id(a)] = (a, a_ptr)
    return a


_I3P_CACHE = {}


def i3p(a):
    hit = _I3P_CACHE.get(id(a))
    if hit is not None and hit[0] is a:
        return hit[1]
    return a.ctypes.data_as(c_i32p)


def f32p(a):
    return a.ctypes.data_as(c_f32p)


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
