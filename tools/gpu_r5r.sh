#!/bin/bash
# rulebook kernels of the in-order step under the profiler (timing of rb_mark_multi variants)
out=gpurun_out/r5r; mkdir -p $out
cd /root/repo
timeout 300 python -m pytest tests/test_hip_chain_rulebooks.py -q -m gpu -x > $out/t_$1.txt 2>&1; tail -1 $out/t_$1.txt
export TMPDIR=/tmp
(cd /tmp && BTC_SCHEDULE=in_order timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r5r -o bench -- python /root/repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras > /root/repo/$out/serial_$1.json 2> /root/repo/$out/serial_$1.err)
find /tmp/prof_r5r -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/serial_$1_kernel_stats.csv
python - "$1" <<'PY'
import csv, sys
rows = list(csv.DictReader(open("gpurun_out/r5r/serial_%s_kernel_stats.csv" % sys.argv[1])))
steps = [int(r["Calls"]) for r in rows if "adam_apply" in r["Name"]][0] / 2
tot = [0, 0]
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"])):
    if any(k in r["Name"] for k in ("rb_", "order_local")):
        tot[0] += int(r["Calls"]); tot[1] += int(r["TotalDurationNs"])
        print("   %6.2f /step %7.1f us/step avg %6.1f min %6.1f max %6.1f %s" % (int(r["Calls"]) / steps, int(r["TotalDurationNs"]) / steps / 1e3, int(r["TotalDurationNs"]) / int(r["Calls"]) / 1e3, int(r["MinNs"]) / 1e3, int(r["MaxNs"]) / 1e3, r["Name"][:50]))
print("rulebook family: %.1f launches, %.1f us per step" % (tot[0] / steps, tot[1] / steps / 1e3))
PY
