"""Seeded synthetic KITTI-shaped scenes: there is no network / dataset here, so
bench.py, smoke() and the parity tests all draw their inputs from this generator.

A 64-beam scan (elevations linspace(-24.9, 2.0, 64) deg, azimuth -40.5..40.5 deg step 0.1728 deg); each
ray hits the ground plane z = -1.73 m or one of 40 random vertical occluders; range noise N(0, 0.02 m);
intensity U[0,1]; cropped to x in [0,70.4], y in [-40,40].  GT boxes sit at occluder positions;
``bm_points`` are 500 points/box uniform in the box; ``rot_z`` ~ U[-45,45] deg with
``pre_rot_points`` = points rotated back by -rot_z (the reference's random_world_rotation keeps the
un-rotated copy, /root/reference/btcdet/datasets/augmentor/data_augmentor.py:136-155).
"""
import numpy as np

KITTI_DET_RANGE = np.array([0, -40, -3, 70.4, 40, 1], dtype=np.float32)
KITTI_OCC_RANGE = np.array([2.24, -40.6944, -2.6, 69.12, 40.6944, 0.64], dtype=np.float32)
KITTI_OCC_VOXEL = [0.32, 0.5184, 0.36]
KITTI_DET_VOXEL = [0.05, 0.05, 0.1]
KITTI_SPHERE_RANGE = [2.24, -40.6944, -16.5953125, 70.72, 40.6944, 4.0, 0.4203125]


def _rotz(points_xyz, deg):
    a = np.deg2rad(deg)
    c, s = np.cos(a), np.sin(a)
    R = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]], dtype=np.float64)  # common_utils.rotate_points_along_z
    return (points_xyz.astype(np.float64) @ R).astype(np.float32)


# Waymo-shaped synthetic workload (BASELINE.json configs[4]; the reference ships no Waymo yaml, so the constants are this
# repository's): 360 deg 64-beam scan, ~170 k points, detection grid 1504 x 1504 x 40.
WAYMO_DET_RANGE = np.array([-75.2, -75.2, -2, 75.2, 75.2, 4], dtype=np.float32)
WAYMO_DET_VOXEL = [0.1, 0.1, 0.15]

PROFILES = {
    "kitti": dict(el=(-24.9, 2.0), az=(-40.5, 40.5), az_step=0.1728, sensor_h=1.73, n_occluders=40, occ_r=(8, 70), occ_az=(-0.7, 0.7),
                  occ_top=0.3, crop=(0, -40, 70.4, 40), boxes=(2, 7), box_lwh=(3.9, 1.6, 1.56), rot=45.0),
    "waymo": dict(el=(-17.6, 2.4), az=(-180.0, 180.0), az_step=0.135, sensor_h=1.8, n_occluders=160, occ_r=(6, 74), occ_az=(-np.pi, np.pi),
                  occ_top=1.2, crop=(-75.2, -75.2, 75.2, 75.2), boxes=(12, 33), box_lwh=(4.7, 2.1, 1.7), rot=45.0),
}


def make_scene(seed, az_step=None, n_occluders=None, n_boxes=None, profile="kitti"):
    pf = PROFILES[profile]
    az_step = pf["az_step"] if az_step is None else az_step
    n_occluders = pf["n_occluders"] if n_occluders is None else n_occluders
    rng = np.random.default_rng(seed)
    el = np.deg2rad(np.linspace(pf["el"][0], pf["el"][1], 64))
    az = np.deg2rad(np.arange(pf["az"][0], pf["az"][1], az_step))
    EL, AZ = np.meshgrid(el, az, indexing="ij")
    dx, dy, dz = np.cos(EL) * np.cos(AZ), np.cos(EL) * np.sin(AZ), np.sin(EL)
    sensor_h = pf["sensor_h"]
    with np.errstate(divide="ignore"):
        r_ground = np.where(dz < 0, -sensor_h / dz, np.inf)
    r = np.minimum(r_ground, 120.0)
    occ_r = rng.uniform(pf["occ_r"][0], pf["occ_r"][1], n_occluders)
    occ_az = rng.uniform(pf["occ_az"][0], pf["occ_az"][1], n_occluders)
    occ_hw = rng.uniform(0.02, 0.08, n_occluders)
    for k in range(n_occluders):
        hit = np.abs(AZ - occ_az[k]) < occ_hw[k]
        rk = occ_r[k] / np.maximum(np.cos(EL), 1e-3)
        zk = rk * dz
        hit &= (zk > -sensor_h) & (zk < pf["occ_top"]) & (rk < r)
        r = np.where(hit, rk, r)
    valid = np.isfinite(r) & (r < 119.0)
    r = r + rng.normal(0, 0.02, r.shape)
    pts = np.stack([r * dx, r * dy, r * dz, rng.uniform(0, 1, r.shape)], axis=-1)[valid].astype(np.float32)
    c = pf["crop"]
    m = (pts[:, 0] >= c[0]) & (pts[:, 0] <= c[2]) & (pts[:, 1] >= c[1]) & (pts[:, 1] <= c[3])
    pts = pts[m]
    nb = int(n_boxes if n_boxes is not None else rng.integers(pf["boxes"][0], pf["boxes"][1]))
    ids = rng.choice(n_occluders, nb, replace=False)
    boxes = np.zeros((nb, 8), dtype=np.float32)
    boxes[:, 0] = occ_r[ids] * np.cos(occ_az[ids])  # centred on the occluder face so that scan points fall inside
    boxes[:, 1] = occ_r[ids] * np.sin(occ_az[ids])
    boxes[:, 3:6] = pf["box_lwh"]
    boxes[:, 2] = -sensor_h + 0.5 * pf["box_lwh"][2]
    boxes[:, 6] = rng.uniform(-np.pi, np.pi, nb)
    boxes[:, 7] = 1
    bm = []
    for b in boxes:
        loc = rng.uniform(-0.5, 0.5, (500, 3)) * b[3:6]
        c, s = np.cos(b[6]), np.sin(b[6])
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
        bm.append((loc @ R.T + b[:3]).astype(np.float32))
    bm = np.concatenate(bm, axis=0) if bm else np.zeros((0, 3), np.float32)
    rot_z = float(rng.uniform(-pf["rot"], pf["rot"]))
    perm = rng.permutation(pts.shape[0])  # shuffle_points (data_processor.py:41-51)
    pts = pts[perm]
    pre_rot = pts.copy()
    pre_rot[:, :3] = _rotz(pts[:, :3], -rot_z)
    return {"points": pts, "pre_rot_points": pre_rot, "gt_boxes": boxes, "bm_points": bm, "rot_z": np.float32(rot_z),
            "box_mirr_flag": np.ones((nb,), np.float32)}


def make_batch(seeds, **kw):
    """collate_batch layout (/root/reference/btcdet/datasets/dataset.py:167-223)."""
    scenes = [make_scene(s, **kw) for s in seeds]
    B = len(scenes)
    maxg = max(s["gt_boxes"].shape[0] for s in scenes)
    gt = np.zeros((B, maxg, 8), np.float32)
    mirr = np.zeros((B, maxg), np.float32)
    for i, s in enumerate(scenes):
        g = s["gt_boxes"].shape[0]
        gt[i, :g] = s["gt_boxes"]
        mirr[i, :g] = s["box_mirr_flag"]
    offs = np.cumsum([0] + [s["points"].shape[0] for s in scenes]).astype(np.int32)
    return {
        "batch_size": B,
        "scenes": scenes,
        "points": np.concatenate([np.pad(s["points"], ((0, 0), (1, 0)), constant_values=i) for i, s in enumerate(scenes)]),
        "pre_rot_points": np.concatenate([s["pre_rot_points"] for s in scenes]),
        "scene_offsets": offs,
        "gt_boxes": gt,
        "gt_boxes_num": [int(s["gt_boxes"].shape[0]) for s in scenes],
        "box_mirr_flag": mirr,
        "bm_points": np.concatenate([np.pad(s["bm_points"], ((0, 0), (1, 0)), constant_values=i) for i, s in enumerate(scenes)]),
        "rot_z": np.array([s["rot_z"] for s in scenes], np.float32),
    }
