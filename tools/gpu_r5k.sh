#!/bin/bash
# round 5: conv_wgrad_x with two items of gathered rows in flight: tests, per-layer A/B against one item in flight (tune 16 = 1)
out=gpurun_out/r5k; mkdir -p $out
cd /root/repo
timeout 600 python -m pytest tests/test_hip_wgrad_x.py -q -m gpu -x > $out/t.txt 2>&1; tail -3 $out/t.txt
ALT=20=1 timeout 300 python tools/wgrad_bench.py > $out/wgrad_f32.txt 2>&1; tail -32 $out/wgrad_f32.txt
BF=1 ALT=20=1 timeout 300 python tools/wgrad_bench.py > $out/wgrad_bf16.txt 2>&1; tail -32 $out/wgrad_bf16.txt
