#!/bin/bash
# round 5, second GPU session: bf16-pipe weight gradient (all shapes) + batched reduction, MR = 2 apply instances, where the small torch ops come from
out=gpurun_out/r5b; mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests/test_hip_wgrad_x.py tests/test_hip_defer_wgrad.py -q -s -m gpu > $out/t_wgrad_x.txt 2>&1; tail -8 $out/t_wgrad_x.txt
timeout 600 python -m pytest tests/test_hip_core.py tests/test_hip_bf16.py tests/test_hip_bf16_mfma.py tests/test_hip_row_order.py tests/test_hip_split.py tests/test_hip_prefetch.py -q -m gpu > $out/t_core.txt 2>&1; tail -5 $out/t_core.txt
timeout 300 python tools/wgrad_bench.py > $out/wgrad_f32.txt 2>&1; tail -31 $out/wgrad_f32.txt
BF=1 timeout 300 python tools/wgrad_bench.py > $out/wgrad_bf16.txt 2>&1; tail -31 $out/wgrad_bf16.txt
CB_WARM=24 timeout 400 python tools/conv_bench.py split split:1=9422 split:1=9412 split:1=9424 > $out/conv_mr2.txt 2>&1; tail -60 $out/conv_mr2.txt
AB_STEPS=60 bash tools/ab_env.sh 2 "old:BTC_TUNE=18=1,BTC_WGRAD_BATCH_REDUCE=0" "newk:BTC_WGRAD_BATCH_REDUCE=0" "new:" > $out/ab_f32.txt 2>&1; cat $out/ab_f32.txt
ALL_OPS=1 TOP=150 timeout 300 python tools/op_sites.py > $out/op_sites.txt 2>&1; tail -170 $out/op_sites.txt
