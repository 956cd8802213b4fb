#!/bin/bash
# timing experiment only: wgrad_bench fp32 on the library as built
out=gpurun_out/r5l; mkdir -p $out
cd /root/repo
ALT=16=1 timeout 300 python tools/wgrad_bench.py > $out/wgrad_f32_$1.txt 2>&1; grep -E "totals|16049   16049  27  (256|128|64)|31839   31839|209797  209797  27   32   32" $out/wgrad_f32_$1.txt
