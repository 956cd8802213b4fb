"""Helpers shared by the golden-vector generator and the tests that consume the vectors."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _hash01(n, salt):
    """exact integer-hash uniform in [0,1): identical on every machine (no libm involved)"""
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(salt) * np.uint64(40503)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    return (h.astype(np.float64) / 4294967296.0).astype(np.float32)


def synthetic_head_outputs(B, nz, ny, nx):
    """stand-in for the occupancy head outputs: logits (B,2,nz,ny,nx) in [-3,3), residuals (B,3,...) in [-0.1,0.1)"""
    n = B * nz * ny * nx
    logit = (_hash01(2 * n, 1) * np.float32(6.0) - np.float32(3.0)).reshape(B, 2, nz, ny, nx)
    res = (_hash01(3 * n, 2) * np.float32(0.2) - np.float32(0.1)).reshape(B, 3, nz, ny, nx)
    return np.ascontiguousarray(logit), np.ascontiguousarray(res)


def load(tag="small"):
    return np.load(os.path.join(HERE, "btc_%s.npz" % tag))


def unpack_mask(g, key, shape):
    n = int(np.prod(shape))
    return np.unpackbits(g[key])[:n].reshape(shape).astype(bool)


def unsparse(g, key):
    shape = tuple(g[key + "_shape"])
    a = np.zeros(int(np.prod(shape)), np.float32)
    a[g[key + "_idx"]] = g[key + "_val"]
    return a.reshape(shape)
