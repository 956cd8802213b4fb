#!/bin/bash
# kernel statistics of the Waymo-shaped step in order (every kernel alone): where the rulebook stage's time goes at that size
mkdir -p gpurun_out/waymo_rb
export TMPDIR=/tmp
cd /tmp
BTC_SCHEDULE=in_order timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wrb -o bench -- python /root/repo/bench.py --workload waymo --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > /root/repo/gpurun_out/waymo_rb/bench.json 2> /root/repo/gpurun_out/waymo_rb/bench.err
cd /root/repo
find /tmp/prof_wrb -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/waymo_rb/kernel_stats.csv
find /tmp/prof_wrb -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/rb_trace.py {} 60 > gpurun_out/waymo_rb/rb_trace.txt 2>&1
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/waymo_rb/kernel_stats.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:40]:
    print("%-70s calls %5s  avg %9.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
