"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes/numpy front-end of ``oracle/btc_oracle.c`` plus numpy restatements of the reference's numpy
pre-steps.  Nothing under ``btcdet_amd/`` may import this module; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and only as the checker.

PARITY STATUS of the spconv-side functions (voxelizer, rulebook, sparse conv, max-pool, dense):
**parity unpinned** -- see the header of ``btc_oracle.c``.  The numpy pre-steps and the occupancy
generator restatement (``oracle/occ_oracle.py``) are pinned against golden vectors generated from the
importable reference (``tests/golden/``).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    """Compile liboracle with gcc (oracle/Makefile)."""
    so = os.path.join(_HERE, "libbtc_oracle_fast.so")
    src = os.path.join(_HERE, "btc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return so


def _cpu_has_fma():
    try:
        flags = open("/proc/cpuinfo").read()
        return " avx2" in flags and " fma" in flags
    except OSError:
        return False


def lib():
    """liboracle; the -mfma build when the host CPU has FMA (bit-identical results, explicit fmaf() either way)"""
    global _LIB
    if _LIB is None:
        name = "libbtc_oracle_fast.so" if _cpu_has_fma() and os.environ.get("BTC_ORACLE_PLAIN", "0") != "1" else "libbtc_oracle.so"
        so = os.path.join(_HERE, name)
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
    return _LIB


def _ip(a):
    return a.ctypes.data_as(_i32p)


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _i3(v):
    if np.isscalar(v):
        v = [v, v, v]
    a = np.ascontiguousarray(np.asarray(v, dtype=np.int32).reshape(-1))
    assert a.size == 3
    return a


# ------------------------------------------------------------------ numpy pre-steps (reference, in tree)
def mask_points_by_range(points, limit_range):
    """/root/reference/btcdet/utils/common_utils.py:59-62 -- x,y only (z is NOT tested)."""
    return (points[:, 0] >= limit_range[0]) & (points[:, 0] <= limit_range[3]) \
        & (points[:, 1] >= limit_range[1]) & (points[:, 1] <= limit_range[4])


def absxyz_2_cylinxyz_np(points):
    """/root/reference/btcdet/utils/coords_utils.py:282-292 (rho, atan2(-y,x) in degrees, z, extras)."""
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    xydist = np.linalg.norm(points[:, :2], axis=1)
    cylin_y = np.arctan2(-y, x) * 180. / np.pi
    xyz = np.stack([xydist, cylin_y, z], axis=-1)
    if points.shape[1] > 3:
        return np.concatenate([xyz, points[:, 3:]], axis=-1)
    return xyz


def absxyz_2_spherexyz_np(points):
    """/root/reference/btcdet/utils/coords_utils.py:268-279."""
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    dist = np.linalg.norm(points[:, :3], axis=1)
    xydist = np.linalg.norm(points[:, :2], axis=1)
    sphere = np.stack([dist, np.arctan2(-y, x) * 180. / np.pi, np.arctan2(z, xydist) * 180. / np.pi], axis=-1)
    if points.shape[1] > 3:
        return np.concatenate([sphere, points[:, 3:]], axis=-1)
    return sphere


# ------------------------------------------------------------------ voxelizer
class VoxelGeneratorV2:
    """Restatement of spconv.utils.VoxelGeneratorV2 as constructed at
    /root/reference/btcdet/datasets/processor/data_processor.py:68-73,112-117,165-170."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels):
        self.point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        self.voxel_size = np.array(voxel_size, dtype=np.float32)
        grid = (self.point_cloud_range[3:] - self.point_cloud_range[:3]) / self.voxel_size
        self.grid_size = np.round(grid).astype(np.int64)
        self.max_num_points = int(max_num_points)
        self.max_voxels = int(max_voxels)
        self._scratch = np.full(int(np.prod(self.grid_size)), -1, dtype=np.int32)

    def generate(self, points, max_voxels=None):
        points = np.ascontiguousarray(points, dtype=np.float32)
        n, c = points.shape
        mv = int(max_voxels or self.max_voxels)
        voxels = np.zeros((mv, self.max_num_points, c), dtype=np.float32)
        coords = np.zeros((mv, 3), dtype=np.int32)
        num = np.zeros((mv,), dtype=np.int32)
        grid = np.ascontiguousarray(self.grid_size.astype(np.int32))
        m = lib().orc_voxelize(_fp(points), n, c, c, _fp(self.point_cloud_range), _fp(self.voxel_size),
                               _ip(grid), self.max_num_points, mv, _fp(voxels), _ip(coords), _ip(num),
                               _ip(self._scratch))
        return {"voxels": voxels[:m], "coordinates": coords[:m], "num_points_per_voxel": num[:m],
                "voxel_num": m}


# ------------------------------------------------------------------ rulebook
MODE_SUBM, MODE_CONV, MODE_TRANSPOSE = 0, 1, 2


def out_shape(in_shape, k, s, p, d, mode, outpad=0):
    o = np.zeros(3, dtype=np.int32)
    lib().orc_out_shape(_ip(_i3(in_shape)), _ip(_i3(k)), _ip(_i3(s)), _ip(_i3(p)), _ip(_i3(d)),
                        _ip(_i3(outpad)), int(mode), _ip(o))
    return o


def rulebook(indices, in_shape, k, s=1, p=0, d=1, mode=MODE_CONV, outpad=0):
    """-> (out_indices (n_out,4), nbr_out (n_out,K), nbr_in (n_in,K), out_shape)."""
    indices = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1, 4)
    n = indices.shape[0]
    k3, s3, p3, d3 = _i3(k), _i3(s), _i3(p), _i3(d)
    K = int(np.prod(k3))
    osh = out_shape(in_shape, k3, s3, p3, d3, mode, outpad)
    cap = max(n * K, 1) if mode != MODE_SUBM else max(n, 1)
    out_idx = np.zeros((cap, 4), dtype=np.int32)
    nbr_out = np.empty((cap, K), dtype=np.int32)
    nbr_in = np.empty((max(n, 1), K), dtype=np.int32)
    n_out = lib().orc_rulebook(_ip(indices), n, _ip(_i3(in_shape)), _ip(osh), _ip(k3), _ip(s3), _ip(p3),
                               _ip(d3), int(mode), cap, _ip(out_idx), _ip(nbr_out), _ip(nbr_in))
    assert n_out >= 0
    return out_idx[:n_out].copy(), nbr_out[:n_out].copy(), nbr_in[:n].copy(), osh


def canonical_pairs(nbr_out):
    """Canonical rulebook of SURVEY.md App. B.4: per offset k the (in_row, out_row) pairs sorted by
    out_row then in_row.  Returns (pairs list per k, pair_num (K,))."""
    K = nbr_out.shape[1]
    pairs, nums = [], np.zeros(K, dtype=np.int32)
    for k in range(K):
        o = np.nonzero(nbr_out[:, k] >= 0)[0]
        pairs.append(np.stack([nbr_out[o, k], o.astype(np.int32)], axis=0))
        nums[k] = o.size
    return pairs, nums


# ------------------------------------------------------------------ apply
def conv_fwd(feat, W, bias, nbr_out):
    feat = np.ascontiguousarray(feat, dtype=np.float32)
    W = np.ascontiguousarray(W, dtype=np.float32)
    K = nbr_out.shape[1]
    cin, cout = W.shape[-2], W.shape[-1]
    assert W.size == K * cin * cout
    n_out = nbr_out.shape[0]
    out = np.empty((n_out, cout), dtype=np.float32)
    nbr_out = np.ascontiguousarray(nbr_out, dtype=np.int32)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    lib().orc_conv_fwd(_fp(feat), _fp(W), _fp(b) if b is not None else None, _ip(nbr_out), n_out, K, cin,
                       cout, _fp(out))
    return out


def conv_dgrad(dout, W, nbr_in):
    dout = np.ascontiguousarray(dout, dtype=np.float32)
    W = np.ascontiguousarray(W, dtype=np.float32)
    K = nbr_in.shape[1]
    cin, cout = W.shape[-2], W.shape[-1]
    n_in = nbr_in.shape[0]
    din = np.empty((n_in, cin), dtype=np.float32)
    nbr_in = np.ascontiguousarray(nbr_in, dtype=np.int32)
    lib().orc_conv_dgrad(_fp(dout), _fp(W), _ip(nbr_in), n_in, K, cin, cout, _fp(din))
    return din


def conv_wgrad(feat, dout, nbr_out, wshape):
    feat = np.ascontiguousarray(feat, dtype=np.float32)
    dout = np.ascontiguousarray(dout, dtype=np.float32)
    K = nbr_out.shape[1]
    cin, cout = feat.shape[1], dout.shape[1]
    dW = np.empty((K, cin, cout), dtype=np.float32)
    nbr_out = np.ascontiguousarray(nbr_out, dtype=np.int32)
    lib().orc_conv_wgrad(_fp(feat), _fp(dout), _ip(nbr_out), nbr_out.shape[0], K, cin, cout, _fp(dW))
    return dW.reshape(wshape)


def maxpool_fwd(feat, nbr_out):
    feat = np.ascontiguousarray(feat, dtype=np.float32)
    nbr_out = np.ascontiguousarray(nbr_out, dtype=np.int32)
    out = np.empty((nbr_out.shape[0], feat.shape[1]), dtype=np.float32)
    lib().orc_maxpool_fwd(_fp(feat), _ip(nbr_out), nbr_out.shape[0], nbr_out.shape[1], feat.shape[1], _fp(out))
    return out


def maxpool_bwd(feat, out, dout, nbr_in):
    feat = np.ascontiguousarray(feat, dtype=np.float32)
    out = np.ascontiguousarray(out, dtype=np.float32)
    dout = np.ascontiguousarray(dout, dtype=np.float32)
    nbr_in = np.ascontiguousarray(nbr_in, dtype=np.int32)
    din = np.empty_like(feat)
    lib().orc_maxpool_bwd(_fp(feat), _fp(out), _fp(dout), _ip(nbr_in), nbr_in.shape[0], nbr_in.shape[1],
                          feat.shape[1], _fp(din))
    return din


def dense(feat, indices, batch_size, shape):
    feat = np.ascontiguousarray(feat, dtype=np.float32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    sh = _i3(shape)
    out = np.empty((batch_size, feat.shape[1], int(sh[0]), int(sh[1]), int(sh[2])), dtype=np.float32)
    lib().orc_dense(_fp(feat), _ip(indices), feat.shape[0], feat.shape[1], batch_size, _ip(sh), _fp(out))
    return out


# ------------------------------------------------------------------ PassOccVox re-voxelization
def revoxelize(points, coords):
    """Restatement of combine_gt_occ_voxel_point + voxelize_pad
    (/root/reference/btcdet/models/occ_pnt/add_occ_template.py:248-268): unique cells sorted
    lexicographically in (b,z,y,x), points of a cell in input order (stable), zero padded to the
    largest cell.  Returns voxels (M,Pmax,C) f32, num (M,) i64, vcoords (M,4) i64."""
    coords = np.asarray(coords, dtype=np.int64)
    points = np.asarray(points, dtype=np.float32)
    if coords.shape[0] == 0:
        return np.zeros((0, 0, points.shape[1]), np.float32), np.zeros((0,), np.int64), np.zeros((0, 4), np.int64)
    uniq, inv, cnt = np.unique(coords, axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    order = np.argsort(inv, kind="stable")
    start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    slot = np.arange(coords.shape[0]) - start[inv[order]]
    vox = np.zeros((uniq.shape[0], int(cnt.max()), points.shape[1]), np.float32)
    vox[inv[order], slot] = points[order]
    return vox, cnt.astype(np.int64), uniq.astype(np.int64)


def bf16_round(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32: the storage rounding of "bf16 features"
    (BASELINE.json configs[2]); NaN stays NaN"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    out = r.view(np.float32).copy()
    out[np.isnan(x)] = np.nan
    return out.reshape(x.shape)


# ---- rotated BEV overlap / IoU / NMS (parity unpinned, see btc_oracle.c) ----
def boxes_overlap_bev(boxes_a, boxes_b, iou=False):
    a = np.ascontiguousarray(boxes_a, dtype=np.float32).reshape(-1, 7)
    b = np.ascontiguousarray(boxes_b, dtype=np.float32).reshape(-1, 7)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    lib().orc_boxes_pairwise_bev(_fp(a), a.shape[0], _fp(b), b.shape[0], int(bool(iou)), _fp(out))
    return out


def boxes_iou_bev(boxes_a, boxes_b):
    return boxes_overlap_bev(boxes_a, boxes_b, iou=True)


def boxes_iou3d(boxes_a, boxes_b):
    """iou3d_nms_utils.boxes_iou3d_gpu:48-78: BEV overlap x height overlap over the union volume"""
    a = np.asarray(boxes_a, dtype=np.float32).reshape(-1, 7)
    b = np.asarray(boxes_b, dtype=np.float32).reshape(-1, 7)
    ov = boxes_overlap_bev(a, b)
    hmax = np.minimum((a[:, 2] + a[:, 5] / 2)[:, None], (b[:, 2] + b[:, 5] / 2)[None, :])
    hmin = np.maximum((a[:, 2] - a[:, 5] / 2)[:, None], (b[:, 2] - b[:, 5] / 2)[None, :])
    o3 = ov * np.clip(hmax - hmin, 0, None)
    va, vb = (a[:, 3] * a[:, 4] * a[:, 5])[:, None], (b[:, 3] * b[:, 4] * b[:, 5])[None, :]
    return (o3 / np.clip(va + vb - o3, 1e-6, None)).astype(np.float32)


def nms(boxes, scores, thresh, pre_maxsize=None, rotated=True):
    """iou3d_nms_utils.nms_gpu / nms_normal_gpu :81-115: stable descending sort, greedy suppression; returns kept ORIGINAL indices"""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 7)
    order = np.argsort(-np.asarray(scores, dtype=np.float32), kind="stable")
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    sb = np.ascontiguousarray(boxes[order])
    keep = np.zeros((max(sb.shape[0], 1),), dtype=np.int64)
    L = lib()
    L.orc_nms.restype = ctypes.c_int
    n = L.orc_nms(_fp(sb), sb.shape[0], ctypes.c_float(thresh), int(bool(rotated)), keep.ctypes.data_as(ctypes.c_void_p))
    return order[keep[:n]]


# ------------------------------------------------------------------ pointnet2_stack (btc_oracle.c)
def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def ball_query(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
    """pointnet2_utils.BallQuery.forward:11-40 -> (idx (M,nsample) int32 with empty balls zeroed, empty_ball_mask (M,) bool);
    radius: float, or [inner, outer] for the shell query"""
    xyz, new_xyz, cnt, ncnt = _f32(xyz).reshape(-1, 3), _f32(new_xyz).reshape(-1, 3), _i32(xyz_batch_cnt), _i32(new_xyz_batch_cnt)
    M = new_xyz.shape[0]
    idx = np.zeros((M, int(nsample)), dtype=np.int32)
    inner, outer = (float(radius[0]), float(radius[1])) if isinstance(radius, (list, tuple)) else (-1.0, float(radius))
    lib().orc_ball_query(_fp(new_xyz), _ip(ncnt), _fp(xyz), _ip(cnt), int(cnt.shape[0]), M, ctypes.c_float(inner), ctypes.c_float(outer),
                         int(nsample), _ip(idx))
    empty = idx[:, 0] == -1
    idx[empty] = 0
    return idx, empty


def group_points(features, features_batch_cnt, idx, idx_batch_cnt):
    """GroupingOperation.forward:52-84 -> (M, C, nsample)"""
    f, fc, idx, ic = _f32(features), _i32(features_batch_cnt), _i32(idx), _i32(idx_batch_cnt)
    M, ns = idx.shape
    out = np.empty((M, f.shape[1], ns), dtype=np.float32)
    lib().orc_group_points(_fp(f), _ip(fc), _ip(idx), _ip(ic), int(ic.shape[0]), M, int(f.shape[1]), ns, _fp(out))
    return out


def group_points_grad(grad_out, idx, idx_batch_cnt, features_batch_cnt, N):
    """GroupingOperation.backward:86-104 -> (N, C)"""
    g, idx, ic, fc = _f32(grad_out), _i32(idx), _i32(idx_batch_cnt), _i32(features_batch_cnt)
    M, C, ns = g.shape
    out = np.empty((int(N), C), dtype=np.float32)
    lib().orc_group_points_grad(_fp(g), _ip(idx), _ip(ic), _ip(fc), int(ic.shape[0]), M, C, int(N), ns, _fp(out))
    return out


def furthest_point_sample(xyz, npoint):
    """FurthestPointSampling.forward:194-213: xyz (B,N,3) -> (B,npoint) int32"""
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    temp = np.full((B, N), 1e10, dtype=np.float32)
    out = np.zeros((B, int(npoint)), dtype=np.int32)
    lib().orc_furthest_point_sampling(_fp(xyz), B, N, int(npoint), _fp(temp), _ip(out))
    return out


def three_nn(unknown, unknown_batch_cnt, known, known_batch_cnt):
    """ThreeNN.forward:222-247 -> (dist (N,3) = sqrt of the squared distances, idx (N,3) global rows)"""
    u, uc, k, kc = _f32(unknown).reshape(-1, 3), _i32(unknown_batch_cnt), _f32(known).reshape(-1, 3), _i32(known_batch_cnt)
    N = u.shape[0]
    d2 = np.zeros((N, 3), dtype=np.float32)
    idx = np.zeros((N, 3), dtype=np.int32)
    lib().orc_three_nn(_fp(u), _ip(uc), _fp(k), _ip(kc), int(uc.shape[0]), N, _fp(d2), _ip(idx))
    with np.errstate(invalid="ignore"):
        return np.sqrt(d2), idx


def three_interpolate(features, idx, weight):
    f, idx, w = _f32(features), _i32(idx), _f32(weight)
    out = np.empty((idx.shape[0], f.shape[1]), dtype=np.float32)
    lib().orc_three_interpolate(_fp(f), _ip(idx), _fp(w), int(idx.shape[0]), int(f.shape[1]), _fp(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, M):
    g, idx, w = _f32(grad_out), _i32(idx), _f32(weight)
    out = np.empty((int(M), g.shape[1]), dtype=np.float32)
    lib().orc_three_interpolate_grad(_fp(g), _ip(idx), _fp(w), int(g.shape[0]), int(g.shape[1]), int(M), _fp(out))
    return out
