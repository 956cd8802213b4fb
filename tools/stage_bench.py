"""Bandwidth of the pre-backbone stages away from the launch-latency floor (SURVEY.md §8d "consequence to state up front"): the
voxelizer, the submanifold rulebook and the strided rulebook on active sets from KITTI size to 100x KITTI size, as algorithmic
bytes (SURVEY §8d formulas) per second.  At KITTI size these stages move 2-5 MB per scene and a launch costs more than the
traffic; this shows what the same kernels reach when the traffic dominates."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from btcdet_amd.spconv import ops, utils

dev = torch.device("cuda:0")
ops.LOOKAHEAD = False if hasattr(ops, "LOOKAHEAD") else None


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


res = {}
grid = (40, 1504, 1504)                     # the Waymo-shaped detection grid [z, y, x]
rng = np.random.default_rng(0)
for n in (30000, 300000, 3000000):
    # active cells: a thin ground-like slab, as LiDAR scenes are (neighbours exist in-plane, few across z)
    lin = rng.choice(grid[1] * grid[2] * 4, size=n, replace=False)
    z, rem = lin // (grid[1] * grid[2]) + 10, lin % (grid[1] * grid[2])
    idx = np.stack([np.zeros_like(z), z, rem // grid[2], rem % grid[2]], axis=1).astype(np.int32)
    I = torch.from_numpy(idx).to(dev)
    out = {}
    for name, kw in (("subm_k3", dict(ksize=3, stride=1, padding=1, subm=True)), ("conv_k3_s2", dict(ksize=3, stride=2, padding=1, subm=False))):
        rb = ops.build_rulebook(I, 1, list(grid), kw["ksize"], kw["stride"], kw["padding"], 1, 0, kw["subm"], False)
        pairs = int((rb.nbr_out >= 0).sum())
        alg = 16 * rb.n_in + 16 * rb.n_out + 8 * pairs
        t = timed(lambda: ops.build_rulebook(I, 1, list(grid), kw["ksize"], kw["stride"], kw["padding"], 1, 0, kw["subm"], False))
        written = 4 * 27 * (rb.n_in + rb.n_out) + 16 * rb.n_out        # the two dense neighbour maps + output indices actually stored
        out[name] = {"n_out": rb.n_out, "pairs": pairs, "us": round(t * 1e6, 1), "alg_GBps": round(alg / t / 1e9, 1),
                     "stored_map_GBps": round(written / t / 1e9, 1)}
    # a whole encoder chain (subm, s2, subm, s2, subm, s2, subm) on this active set through the chain entry points: ONE
    # read-back, levels ranked on the device (csrc/rulebook.hip)
    if ops.fast() is not None:
        from btcdet_amd import spconv
        from btcdet_amd.spconv.geometry import GeometryPlan
        convs = [spconv.SubMConv3d(4, 4, 3, padding=1, bias=False, indice_key="s1")]
        for lv in (2, 3, 4):
            convs += [spconv.SparseConv3d(4, 4, 3, stride=2, padding=1, bias=False, indice_key="c%d" % lv),
                      spconv.SubMConv3d(4, 4, 3, padding=1, bias=False, indice_key="s%d" % lv)]
        plan = GeometryPlan(convs, list(grid), 1)
        rbs = plan.run(I, {})
        alg = sum(16 * rb.n_in + 16 * rb.n_out + 8 * int((rb.nbr_out >= 0).sum()) for rb in rbs)
        written = sum(4 * rb.K * (rb.n_in + rb.n_out) + (16 * rb.n_out if rb.mode != 0 else 0) for rb in rbs)
        t = timed(lambda: plan.run(I, {}))
        out["chain_7_layers"] = {"rows_per_level": [rb.n_out for rb in rbs[::2]], "us": round(t * 1e6, 1), "alg_GBps": round(alg / t / 1e9, 1),
                                 "stored_map_GBps": round(written / t / 1e9, 1), "alg_MB": round(alg / 1e6, 1)}
    # voxelizer: 10 points per active cell on average, KITTI-like caps scaled
    pts_n = n * 4
    cell = idx[rng.integers(0, n, pts_n)]
    vs = np.array([0.1, 0.1, 0.15], np.float32)
    lo = np.array([-75.2, -75.2, -2.0], np.float32)
    xyz = (cell[:, [3, 2, 1]] + rng.random((pts_n, 3))).astype(np.float32) * vs + lo
    pts = np.concatenate([xyz, rng.random((pts_n, 1)).astype(np.float32)], axis=1)
    P, offs = torch.from_numpy(pts).to(dev), torch.tensor([0, pts_n], dtype=torch.int32, device=dev)
    gen = utils.VoxelGeneratorV2([0.1, 0.1, 0.15], [-75.2, -75.2, -2, 75.2, 75.2, 4], 5, max_voxels=n)
    v, c, num = gen.generate_batch(P, offs)
    M = int(v.shape[0])
    alg = 16 * pts_n + 4 * 4 * 5 * M + 16 * M + 4 * M
    t = timed(lambda: gen.generate_batch(P, offs))
    out["voxelize_P5"] = {"points": pts_n, "voxels": M, "us": round(t * 1e6, 1), "alg_GBps": round(alg / t / 1e9, 1)}
    res["n_active_%d" % n] = out
print(json.dumps(res))
