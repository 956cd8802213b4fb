"""TEST / BASELINE INFRASTRUCTURE ONLY (run by bench.py's cpu_baseline leg): the reference's CPU dataset / voxelize path
(DataProcessor: cylinder transform + occupancy-grid voxelization + detection-grid voxelization,
/root/reference/btcdet/datasets/processor/data_processor.py:105-190) with the C oracle, one process per host core -- the
way the reference runs it in DataLoader workers (tools/train.py:27 --workers).

    python oracle/cpu_voxel_bench.py SEED SECONDS     -> prints "<scenes> <elapsed seconds>"
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(seed, seconds):
    sys.path.insert(0, ROOT)
    from btcdet_amd import synth
    from oracle import oracle as orc
    occ_gen = orc.VoxelGeneratorV2(synth.KITTI_OCC_VOXEL, synth.KITTI_OCC_RANGE, 12, 20000)
    det_gen = orc.VoxelGeneratorV2(synth.KITTI_DET_VOXEL, synth.KITTI_DET_RANGE, 5, 16000)
    s = synth.make_scene(seed)
    n, t0 = 0, time.perf_counter()
    while True:
        cyl = orc.absxyz_2_cylinxyz_np(s["pre_rot_points"])
        r = occ_gen.generate(cyl)
        r["voxels"][..., 1] -= s["rot_z"]
        det_gen.generate(s["points"])
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return n, dt


def all_cores(n_procs, seconds=2.0, timeout=120.0):
    """aggregate scenes/s of n_procs independent worker processes (plain subprocesses: the caller may hold a HIP context)"""
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(7000 + i), str(seconds)], stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, env=env) for i in range(n_procs)]
    deadline = time.time() + timeout
    rate, done = 0.0, 0
    for p in procs:
        try:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
            n, dt = out.decode().split()
            rate += int(n) / float(dt)
            done += 1
        except Exception:
            p.kill()
    return rate, done


if __name__ == "__main__":
    n, dt = worker(int(sys.argv[1]), float(sys.argv[2]))
    print(n, dt)
