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


_SIGS = {
    # name: (restype, argtypes)
    "btc_last_error": (ctypes.c_char_p, []),
    "btc_version": (ci, []),
    "btc_voxelize_ws_bytes": (sz, [ci, ci, ci]),
    "btc_voxelize": (ci, [vp, ci, ci, ci, ci, ci, vp, ci, c_f32p, c_f32p, c_i32p, ci, ci, vp, vp, vp, vp, vp, sz, vp]),
    "btc_cart_to_occ_coords": (ci, [vp, vp, ci, ci, ci, vp]),
    "btc_voxel_shift_col": (ci, [vp, vp, ci, ci, ci, ci, vp, ctypes.c_float, vp]),
    "btc_out_shape": (ci, [c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, ci, c_i32p]),
    "btc_rulebook_subm_ws_bytes": (sz, [ci]),
    "btc_rulebook_subm": (ci, [vp, ci, ci, c_i32p, c_i32p, c_i32p, vp, vp, vp, sz, vp]),
    "btc_rulebook_conv_ws_bytes": (sz, [ci, c_i32p]),
    "btc_rulebook_conv_count": (ci, [vp, ci, ci, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, ci, vp, vp, sz, vp]),
    "btc_rulebook_conv_fill": (ci, [vp, ci, ci, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, ci, ci, vp, vp, vp, vp, sz, vp]),
    "btc_pairs_from_nbr": (ci, [vp, ci, ci, ci, vp, vp, vp]),
    "btc_conv_fwd": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp]),
    "btc_conv_dgrad": (ci, [vp, vp, vp, ci, ci, ci, ci, vp, vp]),
    "btc_conv_wgrad_ws_bytes": (sz, [ci, ci, ci, ci]),
    "btc_conv_wgrad": (ci, [vp, vp, vp, ci, ci, ci, ci, vp, vp, sz, vp]),
    "btc_maxpool_fwd": (ci, [vp, vp, ci, ci, ci, vp, vp]),
    "btc_maxpool_bwd": (ci, [vp, vp, vp, vp, ci, ci, ci, vp, vp]),
    "btc_dense_fwd": (ci, [vp, vp, ci, ci, c_i32p, vp, vp]),
    "btc_dense_bwd": (ci, [vp, vp, ci, ci, c_i32p, vp, vp]),
    "btc_revoxelize_ws_bytes": (sz, [ci, ci, c_i32p]),
    "btc_revoxelize_count": (ci, [vp, ci, ci, c_i32p, vp, vp, vp, sz, vp]),
    "btc_revoxelize_fill": (ci, [vp, vp, ci, ci, ci, c_i32p, ci, ci, vp, vp, vp, vp, sz, vp]),
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
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().btc_last_error().decode("utf-8", "replace")
        raise BtcHipError(f"{what} failed (code {rc}): {msg}")


def stream_ptr():
    return vp(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return vp(0)
    if not t.is_cuda:
        raise BtcHipError("expected a tensor on the GPU (HIP) device")
    if not t.is_contiguous():
        raise BtcHipError("expected a contiguous tensor")
    return vp(t.data_ptr())


def i3(v):
    """int / list / tuple / ndarray -> ctypes int32[3] (accepts the reference's int|list|tuple kernel args)."""
    if isinstance(v, (int, np.integer)):
        v = [int(v)] * 3
    a = np.ascontiguousarray(np.asarray(v).reshape(-1).astype(np.int32))
    if a.size != 3:
        raise BtcHipError(f"expected 3 values, got {v!r}")
    return a


def i3p(a):
    return a.ctypes.data_as(c_i32p)


def f32p(a):
    return a.ctypes.data_as(c_f32p)


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
