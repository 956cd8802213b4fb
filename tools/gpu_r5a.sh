#!/bin/bash
# round 5, first GPU session: the transposing-read probe, the bf16-pipe weight gradient's tests and timings, MR = 2 apply instances
out=gpurun_out/r5a; mkdir -p $out
cd /root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 tools/probes/tr_probe.hip -o /tmp/tr_probe 2>/dev/null && /tmp/tr_probe > $out/tr_probe.txt 2>&1; head -3 $out/tr_probe.txt
timeout 600 python -m pytest tests/test_hip_wgrad_x.py -q -s -m gpu > $out/t_wgrad_x.txt 2>&1; tail -15 $out/t_wgrad_x.txt
timeout 600 python -m pytest tests/test_hip_core.py tests/test_hip_defer_wgrad.py tests/test_hip_bf16.py tests/test_hip_bf16_mfma.py tests/test_hip_row_order.py tests/test_hip_split.py -q -m gpu > $out/t_core.txt 2>&1; tail -5 $out/t_core.txt
timeout 300 python tools/wgrad_bench.py > $out/wgrad_f32.txt 2>&1; tail -32 $out/wgrad_f32.txt
BF=1 timeout 300 python tools/wgrad_bench.py > $out/wgrad_bf16.txt 2>&1; tail -32 $out/wgrad_bf16.txt
CB_WARM=24 timeout 400 python tools/conv_bench.py split split:1=9422 split:1=9412 split:1=9424 > $out/conv_mr2.txt 2>&1; tail -70 $out/conv_mr2.txt
AB_STEPS=60 bash tools/ab_env.sh 2 "old:BTC_TUNE=18=1,BTC_WGRAD_BATCH_REDUCE=0" "new:" > $out/ab_f32.txt 2>&1; cat $out/ab_f32.txt
