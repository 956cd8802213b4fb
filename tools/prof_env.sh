#!/bin/bash
tag=$1; shift
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
(cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o bench -- python /root/repo/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > /root/repo/gpurun_out/$tag/bench_prof.json 2> /root/repo/gpurun_out/$tag/bench_prof.err)
find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/stream_timeline.py {} > gpurun_out/$tag/timeline.txt 2>&1
find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$tag/kernel_stats.csv
head -c 1500 gpurun_out/$tag/timeline.txt
