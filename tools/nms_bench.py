"""Timing of the rotated NMS / IoU kernels at the reference's configured sizes (ROI_HEAD.NMS_CONFIG of
tools/cfgs/model_configs/btcdet_kitti_car.yaml: train 9000 -> 512 @ 0.8, test 1024 -> 100 @ 0.7) with the CPU oracle beside it."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from btcdet_amd import iou3d_nms
from oracle import oracle as orc
from test_oracle_iou3d import rand_boxes

dev = "cuda:0"
res = {}
for tag, n, thresh, spread in (("train_9000@0.8", 9000, 0.8, 35.0), ("test_1024@0.7", 1024, 0.7, 35.0)):
    rng = np.random.default_rng(n)
    # proposals cluster around objects: 150 centres, jittered copies
    centres = rand_boxes(rng, 150, spread)
    boxes = centres[rng.integers(0, 150, n)].copy()
    boxes[:, :2] += rng.normal(0, 0.3, (n, 2)); boxes[:, 6] += rng.normal(0, 0.1, n)
    boxes = boxes.astype(np.float32)
    scores = rng.permutation(n).astype(np.float32)
    b, s = torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev)
    for _ in range(3):
        keep, _ = iou3d_nms.nms_gpu(b, s, thresh)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        keep, _ = iou3d_nms.nms_gpu(b, s, thresh)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    ref = orc.nms(boxes, scores, thresh)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    same = bool(np.array_equal(keep.cpu().numpy(), ref))
    res[tag] = {"gpu_ms_incl_sort_and_count_readback": round(gpu_ms, 3), "cpu_oracle_ms_1_core": round(cpu_ms, 1), "kept": int(keep.numel()),
                "same_kept_set_as_oracle": same, "mask_pairs_upper_triangle": n * (n - 1) // 2}
a = torch.from_numpy(rand_boxes(np.random.default_rng(0), 512, 30.0)).to(dev)
g = torch.from_numpy(rand_boxes(np.random.default_rng(1), 64, 30.0)).to(dev)
for _ in range(3):
    iou3d_nms.boxes_iou3d_gpu(a, g)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    iou3d_nms.boxes_iou3d_gpu(a, g)
torch.cuda.synchronize()
res["iou3d_512x64"] = {"gpu_us": round((time.perf_counter() - t0) / 50 * 1e6, 1)}
print(json.dumps(res))
