#!/bin/bash
# round 3, GPU session 1: the refactored step (trainer in the package, reducer on its own RCCL communicator) on hardware
mkdir -p gpurun_out/s1
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_dist_onegpu.py tests/test_hip_rccl.py tests/test_hip_prefetch.py tests/test_hip_pipeline.py tests/test_hip_dense_head.py -x -q > gpurun_out/s1/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s1/pytest.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/s1/bench_default.json 2> gpurun_out/s1/bench_default.err
BTC_BENCH_FORCE_DIST=1 BTC_SYNC_TIMING=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --no-cpu-baseline --no-roofline --no-extras > gpurun_out/s1/bench_dist1.json 2> gpurun_out/s1/bench_dist1.err
BTC_BENCH_FORCE_DIST=1 BTC_SYNC_TRANSPORT=torch timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --no-cpu-baseline --no-roofline --no-extras > gpurun_out/s1/bench_dist1_torch.json 2> gpurun_out/s1/bench_dist1_torch.err
BTC_BENCH_FORCE_DIST=1 BTC_SCHEDULE=split timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --no-cpu-baseline --no-roofline --no-extras > gpurun_out/s1/bench_dist1_split.json 2> gpurun_out/s1/bench_dist1_split.err
timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras > gpurun_out/s1/bench_default2.json 2> gpurun_out/s1/bench_default2.err
timeout 300 python tools/straggler.py 64 gpurun_out/s1/r03_straggler.json > gpurun_out/s1/straggler.log 2>&1
tail -3 gpurun_out/s1/pytest.log; cat gpurun_out/s1/bench_default.json | cut -c1-600; cat gpurun_out/s1/bench_dist1.json | cut -c1-300; cat gpurun_out/s1/bench_dist1_torch.json | cut -c1-300; cat gpurun_out/s1/bench_dist1_split.json | cut -c1-300; cat gpurun_out/s1/bench_default2.json | cut -c1-300; tail -2 gpurun_out/s1/straggler.log
