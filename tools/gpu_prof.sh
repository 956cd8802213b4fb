#!/bin/bash
# bench line + rocprofv3 kernel statistics of the same command (default schedule and the in-order, one-stream schedule)
tag=${1:-r03a}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
cd /root/repo
[ -n "$SKIP_BENCH" ] || timeout 300 python bench.py --no-cpu-baseline > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-roofline --no-extras > /root/repo/gpurun_out/$tag/bench_prof.json 2> /root/repo/gpurun_out/$tag/bench_prof.err)
find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$tag/bench_kernel_stats.csv
find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/stream_timeline.py {} > gpurun_out/$tag/bench_timeline.txt 2>&1
# (in order, one host thread; the weight gradients keep the trainer's default: deferred to the side stream, their slab reductions batched)
(cd /tmp && BTC_SCHEDULE=in_order timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}s -o bench -- python /root/repo/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > /root/repo/gpurun_out/$tag/serial_prof.json 2> /root/repo/gpurun_out/$tag/serial_prof.err)
find /tmp/prof_${tag}s -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$tag/serial_kernel_stats.csv
find /tmp/prof_${tag}s -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/rb_trace.py {} 36 > gpurun_out/$tag/serial_rb_trace.txt 2>&1
find /tmp/prof_${tag}s -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/step_trace.py {} conv > gpurun_out/$tag/serial_step_conv.txt 2>&1
python - <<PY
import json
for f in ("bench", "bench_prof", "serial_prof"):
    try:
        d = json.loads([l for l in open("gpurun_out/$tag/%s.json" % f) if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d.get("rulebook_hbm_GBps"), (d.get("other_kernels") or {}).get("rulebook"), (d.get("roofline") or {}).get("kernel_ms_per_step"))
    except Exception as e:
        print(f, "failed", e)
PY
