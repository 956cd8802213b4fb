"""Timing of the pointnet2_stack kernels and of the ROI head's micro-scene sparse-conv pyramid at the reference's configured
sizes (btcdet_kitti_car.yaml:260-289: 128 rois per scene, 3^3 grid points -> 6912 queries / micro-scenes per bs=2 batch;
raw-point radii 0.4 / 0.8 / 1.2 / 2.4 with 16 / 16 / 32 / 64 samples), with the C oracle on one host core beside it."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from btcdet_amd import pointnet2_stack as p2, synth
from btcdet_amd.spconv import ops
from oracle import oracle as orc

dev = "cuda:0"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


rng = np.random.default_rng(0)
scenes = synth.make_batch([100, 101])["scenes"]
pts = [s["points"][:, :3].astype(np.float32) for s in scenes]
xyz, cnt = np.concatenate(pts), np.array([p.shape[0] for p in pts], np.int32)
rois = np.concatenate([p[rng.integers(0, p.shape[0], 128)] for p in pts])                     # 256 roi centres on the points
grid = (rois[:, None, :] + rng.uniform(-1.5, 1.5, (256, 27, 3))).reshape(-1, 3).astype(np.float32)
ncnt = np.array([128 * 27, 128 * 27], np.int32)
res = {"points_per_scene": cnt.tolist(), "queries": int(grid.shape[0])}
X, C, Q, QC = t(xyz), t(cnt), t(grid), t(ncnt)
for radius, ns in ((0.4, 16), (0.8, 16), (1.2, 32), (2.4, 64)):
    us = timed(lambda: p2.ball_query(radius, ns, X, C, Q, QC))
    t0 = time.perf_counter()
    ridx, rempty = orc.ball_query(radius, ns, xyz, cnt, grid, ncnt)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    idx, empty = p2.ball_query(radius, ns, X, C, Q, QC)
    res["ball_query_r%.1f_n%d" % (radius, ns)] = {"gpu_us": round(us, 1), "cpu_oracle_ms_1_core": round(cpu_ms, 1),
                                                 "same_as_oracle": bool(np.array_equal(idx.cpu().numpy(), ridx)),
                                                 "empty_balls": int(rempty.sum()),
                                                 "distance_tests_brute_force": int(sum(int(c) * int(q) for c, q in zip(cnt, ncnt)))}
feat = t(rng.standard_normal((xyz.shape[0], 16)).astype(np.float32))
idx, _ = p2.ball_query(1.2, 32, X, C, Q, QC)
res["group_points_C16_n32"] = {"gpu_us": round(timed(lambda: p2.grouping_operation(feat, C, idx, QC)), 1)}
fx = np.stack([p[:16384] if p.shape[0] >= 16384 else np.concatenate([p, p[:16384 - p.shape[0]]]) for p in pts]).astype(np.float32)
FX = t(fx)
us = timed(lambda: p2.furthest_point_sample(FX, 2048), reps=5)
t0 = time.perf_counter(); ref = orc.furthest_point_sample(fx, 2048); cpu_ms = (time.perf_counter() - t0) * 1e3
res["fps_16384_to_2048_x2"] = {"gpu_us": round(us, 1), "cpu_oracle_ms_1_core": round(cpu_ms, 1),
                               "same_as_oracle": bool(np.array_equal(p2.furthest_point_sample(FX, 2048).cpu().numpy(), ref))}

# ROI pyramid: 6912 micro-scenes of [2,4,12] cells, three anisotropic 128-channel SparseConv3d, forward + backward
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_hip_roi_microscenes import GRID, LAYERS, micro_scenes
import btcdet_amd.spconv as spconv
from btcdet_amd.backbones_3d import post_act_block
from functools import partial
B = 6912
idx = micro_scenes(rng, B, 0.2)
norm_fn = partial(torch.nn.BatchNorm1d, eps=1e-3, momentum=0.01)
seq = spconv.SparseSequential(*[post_act_block(128, 128, list(LAYERS[i][0]), norm_fn=norm_fn, stride=list(LAYERS[i][1]), padding=list(LAYERS[i][2]),
                                               indice_key='x_combine_spconv%d' % i, conv_type='spconv') for i in range(3)]).to(dev).train()
f = t(rng.standard_normal((idx.shape[0], 128)).astype(np.float32)).requires_grad_(True)
I = t(idx)


def pyramid():
    x = spconv.SparseConvTensor(f, I, list(GRID), B)
    seq(x).dense().pow(2).mean().backward()


res["roi_pyramid_6912x[2,4,12]_128ch_fwd_bwd"] = {"gpu_us": round(timed(pyramid, reps=10), 1), "active_cells": int(idx.shape[0])}
print(json.dumps(res))
