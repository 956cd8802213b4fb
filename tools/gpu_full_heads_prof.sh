#!/bin/bash
# kernel statistics of the step with every head behind the hot path (bench.py --heads full), in order
mkdir -p gpurun_out/full_heads
export TMPDIR=/tmp
cd /tmp
BTC_SCHEDULE=in_order timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fh -o bench -- python /root/repo/bench.py --heads full --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > /root/repo/gpurun_out/full_heads/bench.json 2> /root/repo/gpurun_out/full_heads/bench.err
cd /root/repo
find /tmp/prof_fh -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/full_heads/kernel_stats.csv
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/full_heads/kernel_stats.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.1f" % (tot / 1e6))
for r in rows[:28]:
    print("%-86s calls %5s  avg %9.1f us  total %8.2f ms" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
