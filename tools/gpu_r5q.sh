#!/bin/bash
# composed multi-level marks: rulebook parity tests in both modes, golden / pipeline tests, in-order kernel statistics
out=gpurun_out/r5q; mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests/test_hip_chain_rulebooks.py tests/test_hip_core.py tests/test_hip_golden_full.py tests/test_hip_pipeline.py tests/test_hip_det_backbone.py -q -m gpu -x > $out/t.txt 2>&1; tail -4 $out/t.txt
export TMPDIR=/tmp
(cd /tmp && BTC_SCHEDULE=in_order timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r5q -o bench -- python /root/repo/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > /root/repo/$out/serial.json 2> /root/repo/$out/serial.err)
find /tmp/prof_r5q -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/serial_kernel_stats.csv
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r5q/serial_kernel_stats.csv")))
steps = [int(r["Calls"]) for r in rows if "adam_apply" in r["Name"]][0] / 2
print("launches/step %.1f kernel ms/step %.3f" % (sum(int(r["Calls"]) for r in rows) / steps, sum(int(r["TotalDurationNs"]) for r in rows) / steps / 1e6))
tot = [0, 0]
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"])):
    if any(k in r["Name"] for k in ("rb_", "order_local")):
        tot[0] += int(r["Calls"]); tot[1] += int(r["TotalDurationNs"])
        print("   %6.2f /step %7.1f us/step avg %6.1f  %s" % (int(r["Calls"]) / steps, int(r["TotalDurationNs"]) / steps / 1e3, int(r["TotalDurationNs"]) / int(r["Calls"]) / 1e3, r["Name"][:70]))
print("rulebook family: %.1f launches, %.1f us per step" % (tot[0] / steps, tot[1] / steps / 1e3))
PY
