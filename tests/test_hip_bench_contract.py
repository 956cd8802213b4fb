"""bench.py's output contract (one JSON line on stdout; the fields the driver and the judge read), exercised end to end on the
GPU with a short run: the pipelined schedule, the roofline leg (HIP events per launch) and the CPU-baseline leg all execute."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # exactly one line on stdout
    return json.loads(lines[0])


def test_bench_json_line_contract():
    d = _run("--steps", "4", "--warmup", "2")
    assert d["metric"].startswith("scenes/s fwd+bwd KITTI-Car bs=2/GPU") and d["unit"] == "scenes/s"
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert d["value"] > 50 and abs(d["value"] - 2 * 1000.0 / d["ms_per_step"]) < 0.01 * d["value"]      # bs = 2 scenes per step
    cfg = d["config"]
    assert "workload" in cfg and "model" not in cfg and cfg["global_batch"] == 2 and cfg["parallelism"] == "dp1"
    # the workload is fixed: every step draws unseen scenes (no recycled batches for the optimizer to memorise); level rows are reported
    assert cfg["distinct_batches"] >= 64 + 4 + 1 and cfg["priming_steps"] == 62
    lv = cfg["level_rows"]
    assert lv["det_L0"]["median"] >= lv["det_voxels"]["median"] > 5000 and lv["det_L1"]["min"] > 1000 and lv["occ_out"]["median"] > 50000
    assert lv["det_voxels_after_pass_occ"]["median"] >= lv["det_voxels"]["median"]
    assert cfg["recurring_batches_scenes_per_s"]["scenes_per_s"] > 50
    r = d["roofline"]
    assert (r["traffic"] is None) == (r["traffic_source"] is None)
    # "latency": neither roof within a factor 5 (frac_hbm_measured and frac_mfma < 0.2); the fraction is still quoted against the HBM roof
    assert r["bound"] in ("hbm", "mfma", "latency") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0.05 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 1e6
    assert r["launches_per_step"] == 54.0 and 10 < r["avg_launch_us"] < 500
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "scenes/s" and "sample" in c
    assert d["rulebook_hbm_GBps"] > 1


def test_bench_in_order_schedule_still_runs():
    """BTC_SCHEDULE=in_order: no worker thread, no side-stream preparation -- the plain step the roofline leg uses"""
    env = dict(os.environ, BTC_SCHEDULE="in_order")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-roofline"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    assert d["value"] > 50 and d["config"]["schedule"].startswith("in order, one stream")


def test_bench_distributed_path_at_world_size_one():
    """BTC_BENCH_FORCE_DIST=1: process group over RCCL, bucketed reducer, barriers and max-over-ranks timing at world size 1 -- the code
    path the driver's N > 1 launches take, on the one GPU a test box has (--priming: fewer optimizer steps before the timed region)"""
    env = dict(os.environ, BTC_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--priming", "8", "--no-cpu-baseline",
                        "--no-roofline", "--no-extras"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # RCCL's banner went to stderr
    d = json.loads(lines[0])
    assert d["value"] > 50 and d["n_gpus"] == 1 and d["config"]["priming_steps"] == 7
    assert d["config"]["collective"]["backend"] == "nccl" and d["config"]["collective"]["world_size"] == 1
