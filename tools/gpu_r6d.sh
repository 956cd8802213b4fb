#!/bin/bash
out=gpurun_out/r6d; mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests/test_hip_prefetch.py tests/test_hip_borrow_canary.py tests/test_hip_pipeline.py tests/test_hip_det_backbone.py tests/test_hip_golden_full.py tests/test_hip_dist_onegpu.py -m gpu -x -q 2>&1 | tail -3
b80() { env "$@" timeout 300 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>> $out/bench.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['step_ms']['median'])"; }
echo "default:   $(b80 A=1)"
echo "default:   $(b80 A=1)"
echo "in_order:  $(b80 BTC_SCHEDULE=in_order)"
BTC_TRAINER_TIMING=1 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > $out/timing.json 2> $out/timing.err; grep "trainer host" $out/timing.err; python -c "
import json; d=json.loads([l for l in open('$out/timing.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"
