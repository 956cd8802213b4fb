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


def init_by_name(module):
    """deterministic, name-keyed values for every floating parameter / buffer of a torch module: the same function runs in
    the golden generator on the REFERENCE's module and in the tests on this repository's module (their state_dict keys are
    equal, tests/test_reference_dropin_cpu.py), so weights never have to be stored.  Scales keep activations O(1)."""
    import zlib
    import torch
    with torch.no_grad():
        for name, t in list(module.named_parameters()) + list(module.named_buffers()):
            if t.numel() == 0 or not t.dtype.is_floating_point:
                continue
            u = torch.from_numpy(_hash01(t.numel(), zlib.crc32(name.encode()) & 0xFFFF)).reshape(t.shape)
            if name.endswith("running_var"):
                v = 0.5 + u
            elif name.endswith("running_mean"):
                v = 0.2 * u - 0.1
            elif t.dim() == 1 and name.endswith(".weight"):      # BatchNorm gamma
                v = 0.5 + u
            elif t.dim() == 1:                                    # BatchNorm beta / conv bias
                v = 0.4 * u - 0.2
            else:                                                 # conv weight [k.., Cin, Cout]
                fan = t.numel() / t.shape[-1]
                v = (2.0 * u - 1.0) * float(np.sqrt(3.0 / fan))
            t.copy_(v.to(t.dtype))


def digest(a, n=8192):
    """compact fingerprint of a float array: shape, float64 sum / abs-sum, and a strided sample of <= n elements"""
    a = np.ascontiguousarray(a)
    flat = a.reshape(-1)
    stride = max(1, flat.size // n)
    return {"shape": np.array(a.shape, dtype=np.int64), "sum": np.array(flat.sum(dtype=np.float64)),
            "abssum": np.array(np.abs(flat).sum(dtype=np.float64)), "stride": np.array(stride),
            "sample": flat[::stride][:n].astype(np.float32)}


def put_digest(gold, key, a, n=8192):
    for k, v in digest(a, n).items():
        gold["%s__%s" % (key, k)] = v


def check_digest(g, key, a, rtol, atol, what="", sum_rtol=2e-5):
    """-> (max abs error over the sample, relative error of the abs-sum); asserts shape, sample and sums"""
    a = np.ascontiguousarray(a)
    assert tuple(a.shape) == tuple(int(v) for v in g[key + "__shape"]), (what or key, a.shape, g[key + "__shape"])
    flat = a.reshape(-1)
    stride, ref = int(g[key + "__stride"]), g[key + "__sample"]
    got = flat[::stride][:ref.size].astype(np.float32)
    err = float(np.abs(got - ref).max()) if ref.size else 0.0
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg=what or key)
    s_ref = float(g[key + "__abssum"])
    s_err = abs(float(np.abs(flat).sum(dtype=np.float64)) - s_ref) / max(s_ref, 1e-30)
    assert s_err <= max(10 * rtol, sum_rtol), (what or key, s_err)
    return err, s_err


def sha1(a):
    import hashlib
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8).copy()


def _synth():
    from btcdet_amd import synth
    return synth


def raw_scene(spec):
    """synth scene + points that leave the detection range (so the range mask has work to do); unshuffled order is
    irrelevant -- the reference's shuffle_points permutes `points` (NOT pre_rot_points, data_processor.py:41-51)"""
    s = _synth().make_scene(spec["seed"], n_boxes=spec.get("n_boxes"))
    pts, pre = s["points"], s["pre_rot_points"]
    if spec.get("keep") is not None:
        pts, pre = pts[:spec["keep"]], pre[:spec["keep"]]
    rng = np.random.default_rng(spec["seed"] + 7)
    n_out = 400
    junk = np.stack([rng.uniform(-6, 76, n_out), rng.uniform(-46, 46, n_out), rng.uniform(-2, 0.5, n_out), rng.uniform(0, 1, n_out)], 1).astype(np.float32)
    inside = (junk[:, 0] >= 0) & (junk[:, 0] <= 70.4) & (junk[:, 1] >= -40) & (junk[:, 1] <= 40)
    junk = junk[~inside]
    pos = np.sort(rng.choice(pts.shape[0] + 1, junk.shape[0]))                    # interleave, do not append
    raw = np.insert(pts, pos, junk, axis=0)
    raw_pre = np.insert(pre, pos, np.concatenate([_synth()._rotz(junk[:, :3], -float(s["rot_z"])), junk[:, 3:]], 1), axis=0)
    bm = s["bm_points"] if not spec.get("no_bm") else np.zeros((0, 3), np.float32)
    return s, raw, raw_pre, bm


def make_gt_database(root, n_cars=40, n_peds=12, seed=5):
    """a small ground-truth database in the reference's on-disk format (kitti_dataset.py:296-302: one float32 .bin of
    box-centred points per object + a db_infos dict), written under `root`; deterministic"""
    import pathlib
    root = pathlib.Path(root)
    (root / "gt_database").mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(seed)
    infos = {"Car": [], "Pedestrian": []}
    for name, count, lwh in (("Car", n_cars, (3.9, 1.6, 1.56)), ("Pedestrian", n_peds, (0.8, 0.6, 1.73))):
        for i in range(count):
            box = np.array([rng.uniform(5, 65), rng.uniform(-35, 35), rng.uniform(-1.2, -0.6), lwh[0] * rng.uniform(0.9, 1.1),
                            lwh[1] * rng.uniform(0.9, 1.1), lwh[2] * rng.uniform(0.9, 1.1), rng.uniform(-3.1, 3.1)], dtype=np.float32)
            n = int(rng.integers(3, 60))
            pts = np.concatenate([rng.uniform(-0.5, 0.5, (n, 3)) * box[3:6], rng.uniform(0, 1, (n, 1))], axis=1).astype(np.float32)
            rel = "gt_database/%06d_%s_%d.bin" % (100 + i, name, i % 4)
            pts.tofile(str(root / rel))
            infos[name].append({"name": name, "path": rel, "image_idx": "%06d" % (100 + i), "gt_idx": i % 4, "box3d_lidar": box,
                                "num_points_in_gt": n, "difficulty": int(rng.integers(-1, 3)), "bbox": np.zeros(4, np.float32), "score": -1.0})
    return infos


def variant_inputs(which):
    """inputs of the backbone-variant goldens (gen_variants_golden.py / tests/test_hip_backbone_variants.py), regenerable on both
    sides from numpy + the CPU oracle's voxelizer alone: two synthetic KITTI scenes voxelized on the occupancy grid ("occ": the
    decoder variants, 4 mean features) or on the detection grid ("det": VoxelResBackBone8x, 4 mean features)
    -> (voxel_features (N, 4) f32, voxel_coords (N, 4) int32 [b, z, y, x], grid_size [nx, ny, nz], batch size)"""
    from oracle import oracle as orc
    synth = _synth()
    feats, coords = [], []
    for b, seed in enumerate((41, 42)):
        s = synth.make_scene(seed)
        if which == "occ":
            gen = orc.VoxelGeneratorV2(synth.KITTI_OCC_VOXEL, synth.KITTI_OCC_RANGE, 12, 20000)
            r = gen.generate(orc.absxyz_2_cylinxyz_np(s["pre_rot_points"]))
            grid = [209, 157, 9]
        else:
            gen = orc.VoxelGeneratorV2(synth.KITTI_DET_VOXEL, synth.KITTI_DET_RANGE, 5, 16000)
            r = gen.generate(s["points"])
            grid = [1408, 1600, 40]
        n = np.maximum(r["num_points_per_voxel"], 1).astype(np.float32)[:, None]
        feats.append((r["voxels"].sum(1, dtype=np.float32) / n).astype(np.float32))
        coords.append(np.pad(r["coordinates"], ((0, 0), (1, 0)), constant_values=b).astype(np.int32))
    return np.concatenate(feats), np.concatenate(coords), grid, 2


def convhead_inputs(n_rois=24):
    """inputs of the ROI-head pooling golden (gen_convhead_golden.py / tests/test_hip_conv_head.py), regenerable from numpy + the CPU
    oracle's voxelizer: two synthetic scenes -> raw `points` (N, 5) [b, x, y, z, i], occupancy-completed points `occ_pnts` (M, 4) [x, y,
    z, prob] + their batch index, an `x_combine`-like sparse volume on the stride-8 detection grid [5, 200, 176] (the scenes' detection
    voxels down-sampled by 8, 128 hash-valued channels), and `rois` (2, n_rois, 7): the scenes' boxes jittered by hash values"""
    from oracle import oracle as orc
    synth = _synth()
    pts, occ, occ_b, xc_idx, rois = [], [], [], [], []
    for b, seed in enumerate((51, 52)):
        s = synth.make_scene(seed)
        p = s["points"]
        pts.append(np.concatenate([np.full((p.shape[0], 1), b, np.float32), p], axis=1))
        gen = orc.VoxelGeneratorV2(synth.KITTI_DET_VOXEL, synth.KITTI_DET_RANGE, 5, 16000)
        c = gen.generate(p)["coordinates"]                       # (M, 3) z, y, x on [40, 1600, 1408]
        cell = np.unique(c // 8, axis=0)
        cell = cell[(cell[:, 0] < 5) & (cell[:, 1] < 200) & (cell[:, 2] < 176)]
        xc_idx.append(np.concatenate([np.full((cell.shape[0], 1), b, np.int32), cell.astype(np.int32)], axis=1))
        boxes = s["gt_boxes"][:, :7].astype(np.float32)
        if boxes.shape[0] == 0:
            boxes = np.array([[20, 0, -1, 3.9, 1.6, 1.56, 0.3]], np.float32)
        u = _hash01(n_rois * 7, 100 + b).reshape(n_rois, 7)
        r = boxes[np.arange(n_rois) % boxes.shape[0]].copy()
        r[:, 0:3] += (u[:, 0:3] - np.float32(0.5)) * np.array([2.0, 2.0, 0.4], np.float32)
        r[:, 3:6] *= (np.float32(0.9) + np.float32(0.2) * u[:, 3:6])
        r[:, 6] += (u[:, 6] - np.float32(0.5)) * np.float32(0.6)
        rois.append(r.astype(np.float32))
        # occupancy-completed points: a hash-jittered subset of the box interiors (what PassOccVox adds), probability in the 4th column
        k = 400
        v = _hash01(k * 4, 200 + b).reshape(k, 4)
        bx = boxes[np.arange(k) % boxes.shape[0]]
        o = np.concatenate([bx[:, 0:3] + (v[:, 0:3] - np.float32(0.5)) * bx[:, 3:6], np.float32(0.3) + np.float32(0.7) * v[:, 3:4]], axis=1)
        occ.append(o.astype(np.float32))
        occ_b.append(np.full((k,), b, np.int64))
    xc_idx = np.concatenate(xc_idx)
    xc_feat = (_hash01(xc_idx.shape[0] * 128, 7) - np.float32(0.3)).clip(0).reshape(-1, 128).astype(np.float32)
    return {"points": np.concatenate(pts).astype(np.float32), "occ_pnts": np.concatenate(occ), "added_occ_b_ind": np.concatenate(occ_b),
            "xc_indices": xc_idx, "xc_features": xc_feat, "xc_shape": [5, 200, 176], "rois": np.stack(rois), "batch_size": 2}


def roi_target_inputs(n_rois=512):
    """inputs of the ROI-head target golden (gen_roi_targets_golden.py / tests/test_hip_roi_targets.py): per scene the synthetic scene's
    boxes (B, G + 1, 8) zero-padded with class 1 in the last column, and `n_rois` proposals: a third jittered lightly around the boxes
    (foreground), a third jittered more (hard background), the rest anywhere in the range (easy background); scene 2 has NO boxes
    (the reference's empty-ground-truth path); scores hash-valued, labels 1"""
    synth = _synth()
    boxes = [synth.make_scene(61)["gt_boxes"].astype(np.float32), synth.make_scene(62)["gt_boxes"].astype(np.float32),
             np.zeros((0, 8), np.float32)]
    g = max(len(b) for b in boxes)
    B = len(boxes)
    gt = np.zeros((B, g + 1, 8), np.float32)
    rois = np.zeros((B, n_rois, 7), np.float32)
    for i, bx in enumerate(boxes):
        gt[i, :len(bx)] = bx
        u = _hash01(n_rois * 7, 300 + i).reshape(n_rois, 7)
        base = bx[:, :7] if len(bx) else np.array([[20, 0, -1, 3.9, 1.6, 1.56, 0.3]], np.float32)
        r = base[np.arange(n_rois) % base.shape[0]].copy()
        k = np.arange(n_rois)
        scale = np.where(k % 3 == 0, 0.15, np.where(k % 3 == 1, 1.0, 0.0)).astype(np.float32)[:, None]
        r[:, 0:3] += (u[:, 0:3] - np.float32(0.5)) * np.array([2.0, 2.0, 0.5], np.float32) * scale
        r[:, 3:6] *= (np.float32(1.0) + (u[:, 3:6] - np.float32(0.5)) * np.float32(0.3) * scale)
        r[:, 6] += (u[:, 6] - np.float32(0.5)) * np.float32(0.8) * scale[:, 0]
        far = k % 3 == 2
        r[far, 0] = np.float32(5.0) + u[far, 0] * np.float32(60.0)
        r[far, 1] = (u[far, 1] - np.float32(0.5)) * np.float32(70.0)
        r[far, 6] = (u[far, 6] - np.float32(0.5)) * np.float32(6.0)
        rois[i] = r
    scores = _hash01(B * n_rois, 310).reshape(B, n_rois).astype(np.float32)
    return {"batch_size": B, "rois": rois, "roi_scores": scores, "roi_labels": np.ones((B, n_rois), np.int64), "gt_boxes": gt}
