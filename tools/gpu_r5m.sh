#!/bin/bash
# timing experiment only: conv_bench default policy on the library as built
out=gpurun_out/r5m; mkdir -p $out
cd /root/repo
timeout 300 python tools/conv_bench.py split > $out/conv_$1.txt 2>&1; tail -45 $out/conv_$1.txt
