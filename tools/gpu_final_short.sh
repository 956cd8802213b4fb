#!/bin/bash
# the measurement set of a round when GPU minutes are short: the driver-equivalent bench line, the Waymo-shape line, rocprofv3 kernel
# statistics of the bench command (default and in-order schedule), the two PMC passes -> gpurun_out/<tag>/, most important first
tag=${1:-r04w}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
cd /root/repo
timeout 300 python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
timeout 200 python bench.py --workload waymo --steps 40 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/$tag/bench_waymo.json 2> gpurun_out/$tag/bench_waymo.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-roofline --no-extras > /root/repo/gpurun_out/$tag/bench_prof.json 2> /root/repo/gpurun_out/$tag/bench_prof.err)
find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$tag/bench_kernel_stats.csv
find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/stream_timeline.py {} > gpurun_out/$tag/bench_timeline.txt 2>&1
bash tools/gpu_pmc.sh $tag > gpurun_out/$tag/pmc.log 2>&1
(cd /tmp && BTC_SCHEDULE=in_order BTC_DEFER_WGRAD=0 BTC_OVERLAP_MIN_ROWS=2000000000 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}s -o bench -- python /root/repo/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > /root/repo/gpurun_out/$tag/serial_prof.json 2> /root/repo/gpurun_out/$tag/serial_prof.err)
find /tmp/prof_${tag}s -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$tag/serial_kernel_stats.csv
find /tmp/prof_${tag}s -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/step_trace.py {} conv > gpurun_out/$tag/serial_step_conv.txt 2>&1
for f in bench bench_waymo bench_prof serial_prof; do
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/$tag/$f.json") if l.startswith("{")][-1])
    r = d.get("roofline") or {}
    c = d.get("config") or {}
    print("$f", d["value"], d["ms_per_step"], "conv ms", r.get("kernel_ms_per_step"), "frac", r.get("frac"), "traffic", r.get("traffic"), "in order", c.get("in_order_scenes_per_s"), "rpn", c.get("with_rpn_heads"), "all", c.get("with_all_heads"), "recur", c.get("recurring_batches_scenes_per_s"))
except Exception as e:
    print("$f failed", e)
PY
done
tail -3 gpurun_out/$tag/pmc.log
