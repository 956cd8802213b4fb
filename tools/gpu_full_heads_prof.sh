#!/bin/bash
# steady-state kernel statistics of the step with every head behind the hot path (bench.py --heads full), in order:
# the last 10 timed steps of the trace only (tools/trace_tail.py) -- MIOpen's find runs and the warm-up stay out
mkdir -p gpurun_out/full_heads
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_fh
BTC_SCHEDULE=${SCHED:-in_order} timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_fh -o bench -- python /root/repo/bench.py --heads ${HEADS:-full} --priming ${PRIMING:-24} --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > /root/repo/gpurun_out/full_heads/bench.json 2> /root/repo/gpurun_out/full_heads/bench.err
cd /root/repo
ms=$(python -c "
import json
d=json.loads([l for l in open('gpurun_out/full_heads/bench.json') if l.startswith('{')][-1]); print(d['ms_per_step'])")
# tail window = 8 steps' worth of time, ending at the last kernel (the timed region ends the run: --no-extras)
find /tmp/prof_fh -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/trace_tail.py {} 8 -$(python -c "print(8*$ms)") 60 > gpurun_out/full_heads/tail_${HEADS:-full}.txt 2>&1
head -70 gpurun_out/full_heads/tail_${HEADS:-full}.txt
