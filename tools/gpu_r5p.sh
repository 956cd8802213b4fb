#!/bin/bash
# packed fp32 subtraction in the operand split: parity tests, per-layer timings of both split-operand kernels
out=gpurun_out/r5p; mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests/test_hip_wgrad_x.py tests/test_hip_split.py tests/test_hip_row_order.py -q -m gpu -x > $out/t.txt 2>&1; tail -3 $out/t.txt
ALT=16=2 timeout 300 python tools/wgrad_bench.py > $out/wgrad_f32.txt 2>&1; tail -1 $out/wgrad_f32.txt
timeout 300 python tools/conv_bench.py split > $out/conv_split.txt 2>&1; tail -2 $out/conv_split.txt
